#!/bin/bash
# usage (under gpurun): tools/perf_ab.sh [pytest -k expression] -- quick A/B: bitwise tests, then the 1M bench
mkdir -p gpurun_out
K="${1:-tiled or reproduc or creeping_counterflow_200 or walking or hotspot}"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiles.py -m gpu -q -x -k "$K" > gpurun_out/ab_tests.log 2>&1
rc=$?
tail -4 gpurun_out/ab_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
run() { timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4), 'value %.3g' % r['value'])"; }
echo -n "1M e2c2: "; run || exit 1
echo -n "1M e1c1: "; run --eyesight 1.0 --cell 1.0 || exit 1
echo -n "1M hotspots: "; run --workload hotspots || exit 1
echo -n "1M random: "; run --workload random || exit 1
