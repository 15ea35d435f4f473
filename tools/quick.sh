run() { timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4))"; }
echo -n "default: "; run
