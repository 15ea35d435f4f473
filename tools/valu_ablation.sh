#!/bin/bash
# usage (under gpurun): tools/valu_ablation.sh -- wave instructions per launch of the tiled step kernel with
# parts of it switched off (-DCS_TILE_ABLATION build): where the instructions are.
# bits: 1 = staging + epilogue only, 5 = staging only, 16 = + per-agent setup and empty loops,
#       8 = + filter (no time-to-collision / forces), 0 = everything, 4 = everything but the epilogue
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CS_HIPCC_EXTRA=-DCS_TILE_ABLATION python -c "from rmf_crowdsim_amd import _native; _native.build(force=True)" || exit 1
for d in ${@:-0 5 1 16 8 4}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/valu_abl_$d
  rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out -o p --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-creep-leg --debug $d > $out/out.txt 2> $out/err.txt || { echo pmc failed $d; tail -5 $out/err.txt; exit 1; }
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step_tiled" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
print("debug $d:", "  ".join(f"{c} {agg[c]/cnt[c]:.4g}" for c in sorted(agg)))
PY
  rm -rf $out/*/  # keep only the logs
done
python -c "from rmf_crowdsim_amd import _native; _native.build(force=True)"
