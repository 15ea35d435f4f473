run() { timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4))"; }
for cfg in "--cell 2.0 --eyesight 2.0" "--cell 1.0 --eyesight 1.0"; do
for d in 0 8 16 1 3; do echo -n "$cfg debug $d: "; run $cfg --debug $d; done; done
