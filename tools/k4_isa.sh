#!/bin/bash
# usage: tools/k4_isa.sh [extra hipcc flags] -- compile the engine to /tmp/isa with -save-temps and print
# the step kernel's register use, spill lane traffic and static instruction mix (no GPU needed)
mkdir -p /tmp/isa && cd /tmp/isa && rm -f *.s *.bc *.o *.out *.hipi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -mllvm -amdgpu-sched-strategy=iterative-ilp -fno-unroll-loops -fno-slp-vectorize -gline-tables-only -save-temps -w "$@" \
  -I /root/repo/include -o /tmp/isa/lib.so /root/repo/rmf_crowdsim_amd/csrc/crowdstep_hip.hip 2>&1 | grep -A5 "error" | head -30
S=/tmp/isa/crowdstep_hip-hip-amdgcn-amd-amdhsa-gfx950.s
for k in k_step_tiledILb0 k_step_tiledILb1 k_step_gather; do
python3 /root/repo/tools/isa_by_line.py $S $k | head -1
echo -n "$k: "; awk -v k="$k" '$0 ~ "^_Z[0-9]+"k {f=1} f&&/; (TotalNumSgprs|NumVgprs|ScratchSize|Occupancy|LDSByteSize)/{print} f&&/; LDSByteSize/{exit}' $S | tr '\n' ' '; echo
done
