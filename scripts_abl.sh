run() { timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4))"; }
for b in 5 6 7 8; do echo -n "128thr blocks $b: "; CS_TILE_BLOCKS_PER_CU=$b run; done
echo -n "128 cap40 b7: "; CS_TILE_BLOCKS_PER_CU=7 CS_TILE_LIST_CAP=40 run
echo -n "128 cap40 b8: "; CS_TILE_BLOCKS_PER_CU=8 CS_TILE_LIST_CAP=40 run
