run() { timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4), 'value %.3g' % r['value'])"; }
for t in 224 160 112 64; do echo -n "100k target $t: "; CS_TILE_TARGET=$t run --agents 100000; done
for t in 224 112; do echo -n "300k target $t: "; CS_TILE_TARGET=$t run --agents 300000; done
