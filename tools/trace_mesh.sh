#!/bin/bash
# usage (under gpurun): tools/trace_mesh.sh [agents tiles_x tiles_y] -- the kernels of a tile's step on an in-process mesh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/trace_mesh
rm -rf $out; mkdir -p $out
python3 tools/mesh_steps.py "$@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o t --output-format csv -- python3 tools/mesh_steps.py "$@" > $out/out.txt 2> $out/err.txt
python3 - <<PY
import csv, glob, statistics
f=glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ks={}
for r in rows:
    ks.setdefault(r["Kernel_Name"][:36],[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
for k,v in sorted(ks.items(), key=lambda kv: -sum(kv[1])):
    if len(v)>20: print(f"{k:38s} n {len(v):5d} median {statistics.median(v):7.1f} p90 {sorted(v)[9*len(v)//10]:7.1f} us  total {sum(v)/1000:8.2f} ms")
PY
rm -rf $out/*/
