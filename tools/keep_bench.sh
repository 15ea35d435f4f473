#!/bin/bash
# usage (under gpurun): tools/keep_bench.sh [agents...] -- small crowds: the window builder in every step's own scatter
# launch (CS_WINDOWS_KEEP=0, "shadow 0") against stepping on the windows of the step before, cut by builder workgroups
# inside the neighbour kernel's launch (the default, "shadow 1"), with 5 / 10 / 20 % of room in a window
mkdir -p gpurun_out
sizes=${@:-"125000"}
for w in walk random creep; do
for n in $sizes; do
for cfg in "0 0" "1 5" "1 10" "1 20"; do
  set -- $cfg
  echo -n "$w agents $n shadow $1 slack $2: "
  CS_WINDOWS_KEEP=$1 CS_WINDOWS_SLACK=$2 timeout -k 10 120 python bench.py --workload $w --agents $n --steps 400 --warmup 50 --no-cpu-baseline --no-creep-leg --profile-stride 17 2>gpurun_out/keep.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms/step', round(r['ms_per_step'],4), 'kernel_ms',round(r['roofline']['kernel_ms'],4), 'value %.3g' % r['value'])" || { tail -3 gpurun_out/keep.err; }
done
done
done
