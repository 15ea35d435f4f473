run() { timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('ms/step',round(r['ms_per_step'],4),'k4_ms',round(r['roofline']['kernel_ms'],4))"; }
for b in 2 3 4 5; do echo -n "blocks/CU $b: "; CS_TILE_BLOCKS_PER_CU=$b run; done
for c in 32 40; do echo -n "blocks 4 cap $c: "; CS_TILE_BLOCKS_PER_CU=4 CS_TILE_LIST_CAP=$c run; done
echo -n "cell1 e2: "; run --cell 1.0 --eyesight 2.0
echo -n "cell1 e1: "; run --cell 1.0 --eyesight 1.0
echo -n "debug1: "; run --debug 1
echo -n "debug3: "; run --debug 3
