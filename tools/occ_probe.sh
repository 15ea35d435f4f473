#!/bin/bash
# usage (under gpurun): tools/occ_probe.sh -- does a fifth workgroup per CU pay?  Eyesight 1.5 m (K ~ 18), where 26 list
# rows suffice and five workgroups fit the LDS: the 4-wave build against the 5-wave build (96 VGPRs, spills)
run() { timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-creep-leg --eyesight 1.5 2>gpurun_out/occ.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4))" || tail -3 gpurun_out/occ.err; }
for v in w4 w5; do
  export CS_LIB_PATH=$PWD/rmf_crowdsim_amd/lib/variants/$v.so
  echo -n "$v default cfg: "; CS_TILE_PRINT=1 run; grep -a "tile cfg" gpurun_out/occ.err | head -1
  echo -n "$v list 24, 4 per CU: "; CS_TILE_LIST_CAP=24 CS_TILE_BLOCKS_PER_CU=4 CS_TILE_PRINT=1 run; grep -a "tile cfg" gpurun_out/occ.err | head -1
  echo -n "$v list 24, 5 per CU: "; CS_TILE_LIST_CAP=24 CS_TILE_BLOCKS_PER_CU=5 CS_TILE_PRINT=1 run; grep -a "tile cfg" gpurun_out/occ.err | head -1
done
