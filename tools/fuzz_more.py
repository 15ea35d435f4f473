#!/usr/bin/env python3
"""Further seeds of the randomized mesh tests (tests/test_gpu_tiles.py), by hand: python tools/fuzz_more.py FIRST LAST
Each seed runs with the usual launches and with the border / interior launches (CS_TILE_SPLIT=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import test_gpu_tiles as T  # noqa: E402


class Env:
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


first, last = int(sys.argv[1]), int(sys.argv[2])
ran = skipped = 0
for seed in range(first, last):
    for split in (False, True):
        for fn in (T.test_random_meshes_match_the_single_engine, T.test_random_call_sequences_on_a_mesh_match_the_single_engine):
            os.environ.pop("CS_TILE_SPLIT", None)
            try:
                fn(seed, split, Env())
                ran += 1
            except pytest.skip.Exception:
                skipped += 1
            except Exception:
                print(f"FAILED {fn.__name__} seed {seed} split {split}", flush=True)
                raise
    print(f"seed {seed} ok", flush=True)
print(f"{ran} cases passed, {skipped} skipped")
