#!/usr/bin/env python3
"""How much a tile engine costs over a plain engine on ONE GPU: the same crowd stepped by one
engine, by a 1x1 tile mesh (tile bookkeeping only) and by a 2x1 / 2x2 mesh (halo kernels +
device copies, all tiles sharing the GPU).  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from rmf_crowdsim_amd import LocationHash2D, Simulation, Zanlungo, scenes  # noqa: E402
from rmf_crowdsim_amd.tiles import LocalTileMesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0)
lp = Zanlungo(*scenes.METRIC_ZANLUNGO)


def run(target, sync, steps=100, warm=10):
    scenes.add_counterflow(target, pts, group, scenes.CREEP_SPEED, lp, 2.0)
    for _ in range(warm):
        target.step(0.05, report=False)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        target.step(0.05, report=False)
    sync()
    return (time.perf_counter() - t0) / steps * 1e3


single = Simulation(LocationHash2D(**grid))
print(f"single engine      {run(single, single.synchronize):.3f} ms/step")
del single
for tiles in ((1, 1), (2, 1), (2, 2)):
    mesh = LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=1, density_per_cell=15.0)
    ms = run(mesh, lambda: [e.synchronize() for e in mesh.engines])
    print(f"mesh {tiles}         {ms:.3f} ms/step  ({n // (tiles[0] * tiles[1])} agents per tile)")
    del mesh
