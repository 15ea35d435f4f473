#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags] -- build the engine as it stands into
# rmf_crowdsim_amd/lib/variants/<name>.so (travels with gpurun; timed by tools/variants_bench.sh)
name=$1; shift
mkdir -p /root/repo/rmf_crowdsim_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -mllvm -amdgpu-sched-strategy=iterative-ilp -fno-unroll-loops -fno-slp-vectorize -w "$@" \
  -I /root/repo/include -o /root/repo/rmf_crowdsim_amd/lib/variants/$name.so /root/repo/rmf_crowdsim_amd/csrc/crowdstep_hip.hip && echo built $name
