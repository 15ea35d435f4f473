#!/bin/bash
# usage: tools/abl_variants.sh build            (here: one build of the engine per ablation, compile-time switches)
#        tools/abl_variants.sh run [bench args] (under gpurun: kernel time of each)
# -DCS_TILE_ABL=<bits> compiles the tiled kernel WITHOUT the parts named by the bits (unlike the run-time switches of
# -DCS_TILE_ABLATION, which cost scalar registers and slow the whole kernel down):
#   1 = no neighbour pass (staging + epilogue only), 16 = empty filter, 8 = no time-to-collision (and so no forces),
#   64 = no forces, 32 = the older form of the time-to-collision pass, 0 = the product
BITS=${CS_ABL_BITS:-"0 1 16 8 64 32"}  # 32 = the older time-to-collision pass (per-entry masks) instead of the lean one
if [ "$1" = build ]; then
  for d in $BITS; do bash /root/repo/tools/build_variant.sh abl$d -DCS_TILE_ABL=$d || exit 1; done
  exit 0
fi
shift
mkdir -p gpurun_out
for d in $BITS; do
  so=rmf_crowdsim_amd/lib/variants/abl$d.so
  [ -f $so ] || continue
  echo -n "abl $d: "
  CS_LIB_PATH=$PWD/$so timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-creep-leg --debug $d "$@" 2>gpurun_out/abl.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms', round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4))" || tail -3 gpurun_out/abl.err
done
