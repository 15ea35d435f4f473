#!/bin/bash
# usage (under gpurun): tools/cell_rows_sweep.sh [agents] -- K4 time over LocationHash2D cell sizes (eyesight 2 m fixed)
# and owned rows per band window (CS_TILE_ROWS): review of round 3, item 5 (iii)
agents=${1:-1000000}
mkdir -p gpurun_out
for cell in 2.0 1.0 0.6667 0.5; do
for rows in 2 3 4 6 8; do
  echo -n "agents $agents cell $cell rows $rows: "
  CS_TILE_ROWS=$rows timeout -k 10 120 python bench.py --agents $agents --cell $cell --eyesight 2.0 --steps 100 --warmup 20 --no-cpu-baseline --no-creep-leg 2>gpurun_out/sweep.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4), 'value %.3g' % r['value'])" || { tail -2 gpurun_out/sweep.err; }
done
done
