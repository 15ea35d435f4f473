#!/bin/bash
# usage (under gpurun): tools/ablation.sh -- kernel time of the tiled step kernel with parts switched off
# (profiling build -DCS_TILE_ABLATION; the product library is rebuilt afterwards).
# bits: 1 = no neighbour pass at all (staging + epilogue only), 16 = empty filter (no lists),
#       8 = no time-to-collision (filter only; forces see empty masks), 64 = no forces, 4 = no epilogue,
#       32 = the general form of the neighbour pass instead of the fast one
CS_HIPCC_EXTRA=-DCS_TILE_ABLATION python -c "from rmf_crowdsim_amd import _native; _native.build(force=True)" || exit 1
for d in ${CS_ABLATION_BITS:-0 1 16 8 64 4 32}; do
  echo -n "debug $d: "
  timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --debug $d "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms', round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4))"
done
python -c "from rmf_crowdsim_amd import _native; _native.build(force=True)"
