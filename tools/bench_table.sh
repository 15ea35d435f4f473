#!/bin/bash
# usage (under gpurun): tools/bench_table.sh <tag> -- the workloads of DESIGN.md section 8, one line each
tag=${1:-r03}
mkdir -p gpurun_out/table_$tag
line() { python -c "import json,sys; r=json.loads(sys.stdin.read()); c=r.get('creep_scene'); print('%-34s ms/step %.4f  value %.3g  k4_ms %.4f%s' % (sys.argv[1], r['ms_per_step'], r['value'], r['roofline']['kernel_ms'], ('  creep k4_ms %.4f' % c['kernel_ms']) if c else ''))" "$1"; }
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2> gpurun_out/table_$tag/$name.err | tee gpurun_out/table_$tag/$name.json | line $name || { echo "$name failed"; tail -3 gpurun_out/table_$tag/$name.err; }; }
timeout -k 10 300 python bench.py 2> gpurun_out/table_$tag/default.err | tee gpurun_out/table_$tag/default.json | line "default (walk, cpu baseline)"
run creep --workload creep --no-creep-leg
run random --workload random
run e1c1 --eyesight 1.0 --cell 1.0 --no-creep-leg
run c1e2 --eyesight 2.0 --cell 1.0 --no-creep-leg
run c4e2 --eyesight 2.0 --cell 4.0 --no-creep-leg
run hotspots_1M --workload hotspots
run hotspots_4M --workload hotspots --agents 4000000 --steps 60
run stream --workload stream
run stream_route --workload stream --planner route
run walk_4M --agents 4000000 --steps 100 --no-creep-leg
run walk_16M --agents 16000000 --steps 40 --warmup 5 --no-creep-leg
run walk_125k --agents 125000 --steps 400 --no-creep-leg
run walk_100k --agents 100000 --steps 400 --no-creep-leg
run readback --readback --no-creep-leg --steps 100
