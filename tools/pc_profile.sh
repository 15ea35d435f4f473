#!/bin/bash
# usage (under gpurun): tools/pc_profile.sh <variant> [bench args] -- PC sampling of the bench (rocprofv3 beta);
# the samples are mapped to source lines afterwards by tools/pc_report.py against the same build's code object
v=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export CS_LIB_PATH=$GRAFT_REPO_ROOT/rmf_crowdsim_amd/lib/variants/$v.so
for m in stochastic host_trap; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pcs_$m
  rm -rf $out; mkdir -p $out
  if [ $m = stochastic ]; then unit="cycles"; iv=65536; else unit="time"; iv=1; fi
  timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m --pc-sampling-unit $unit --pc-sampling-interval $iv \
     -d $out -o p --output-format csv -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-creep-leg "$@" > $out/out.txt 2> $out/err.txt
  echo "$m rc=$?"; tail -3 $out/err.txt; find $out -type f | head; 
  for f in $(find $out -name "*pc_sampling*.csv"); do wc -l $f; head -3 $f; gzip -f $f; done
done
