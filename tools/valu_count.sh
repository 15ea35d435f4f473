#!/bin/bash
# usage (under gpurun): tools/valu_count.sh [bench args] -- one PMC pass: wave instructions per launch
# of the tiled step kernel (the kernel is VALU-issue bound: this is the number to bring down)
args=${@:-"--steps 12 --warmup 3 --no-cpu-baseline"}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/valu
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $out -o p --output-format csv -- python3 bench.py $args > /dev/null 2> $out/err.txt || { echo pmc failed; tail -5 $out/err.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step_tiled" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print("k_step_tiled per launch:", "  ".join(f"{c} {agg[c]/cnt[c]:.4g}" for c in sorted(agg)))
PY
