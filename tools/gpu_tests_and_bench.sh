#!/bin/bash
# usage: scripts_gpu.sh <tag> -- helper used with gpurun: tests, then bench variants (skipped after a timeout)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s -x > gpurun_out/tests.log 2>&1
rc=$?
tail -25 gpurun_out/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out; stopping"; exit $rc; fi
for cfg in "2.0 2.0" "1.0 2.0" "1.0 1.0"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --cell $1 --eyesight $2 --no-cpu-baseline > gpurun_out/bench_c$1_e$2.json 2> gpurun_out/bench_c$1_e$2.err
  brc=$?
  cat gpurun_out/bench_c$1_e$2.json; tail -3 gpurun_out/bench_c$1_e$2.err
  if [ $brc -eq 124 ] || [ $brc -eq 137 ]; then echo "bench timed out; stopping"; exit $brc; fi
done
