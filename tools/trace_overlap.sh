cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_overlap
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python3 tools/overlap_ab_one_gpu.py 1000000 > $out/out.txt 2> $out/err.txt
python3 - <<PY
import csv, glob
f=glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find the last 40 kernels of the overlapped run: print a window of consecutive kernels near the end
names=[r["Kernel_Name"][:28] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith("k_wait_border")]
print("wait kernels", len(idx))
i0=idx[len(idx)//2]
t0=int(rows[i0-6]["Start_Timestamp"])
for r in rows[i0-6:i0+14]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1000:9.1f} {(int(r["End_Timestamp"])-t0)/1000:9.1f}  q{r.get("Queue_Id","?")}  {r["Kernel_Name"][:60]}  grid {r["Grid_Size_X"]}')
PY
