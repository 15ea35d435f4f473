run() { timeout -k 10 120 python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4), 'value %.3g' % r['value'])"; }
for n in 10000 100000 300000 1000000 4000000; do echo -n "agents $n: "; run --agents $n; done
