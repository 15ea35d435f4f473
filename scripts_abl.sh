run() { timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('k4_ms',round(r['roofline']['kernel_ms'],4), 'ms/step', round(r['ms_per_step'],4), r['config']['n_nonfinite'], r['config']['n_agents_alive'])"; }
echo -n "rows1: "; CS_TILE_ROWS=1 run
for b in 3 4 5; do echo -n "rows2 b$b: "; CS_TILE_ROWS=2 CS_TILE_BLOCKS_PER_CU=$b run; done
for b in 4 5; do echo -n "rows2 b$b cap40: "; CS_TILE_ROWS=2 CS_TILE_BLOCKS_PER_CU=$b CS_TILE_LIST_CAP=40 run; done
for b in 4 5; do echo -n "rows4 b$b cap40 t200: "; CS_TILE_ROWS=4 CS_TILE_TARGET=200 CS_TILE_BLOCKS_PER_CU=$b CS_TILE_LIST_CAP=40 run; done
echo -n "rows2 t240 cap40: "; CS_TILE_ROWS=2 CS_TILE_TARGET=240 CS_TILE_LIST_CAP=40 run
