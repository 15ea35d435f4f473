#!/bin/bash
# usage (under gpurun): tools/trace_small.sh -- kernel durations of the walking and the standing crowd at 125k agents
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in walk creep; do
  out=$GRAFT_REPO_ROOT/gpurun_out/trace_$w
  rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-creep-leg --agents 125000 --workload $w > $out/out.txt 2> $out/err.txt
  echo "== $w"; python3 - <<PY
import csv, glob, statistics
f=glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ks={}
for r in rows:
    ks.setdefault(r["Kernel_Name"][:24],[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
for k,v in ks.items():
    if len(v)>20: print(f"{k:26s} n {len(v):4d} median {statistics.median(v):7.1f} p10 {sorted(v)[len(v)//10]:7.1f} p90 {sorted(v)[9*len(v)//10]:7.1f} us grid {[r['Grid_Size_X'] for r in rows if r['Kernel_Name'][:24]==k][-1]}")
PY
  rm -rf $out/*/
done
