#!/bin/bash
# usage (under gpurun): tools/variants_bench.sh [bench args] -- the 1M bench for every build under
# rmf_crowdsim_amd/lib/variants/ (CS_LIB_PATH), two rounds so that box drift shows
mkdir -p gpurun_out
for round in 1 2; do
for so in rmf_crowdsim_amd/lib/variants/*.so; do
  echo -n "$round $(basename $so .so): "
  CS_LIB_PATH=$PWD/$so timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-creep-leg "$@" 2>gpurun_out/variant.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4), 'value %.3g' % r['value'])" || { tail -3 gpurun_out/variant.err; }
done
done
