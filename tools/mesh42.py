import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from rmf_crowdsim_amd import LocationHash2D, Simulation, Zanlungo, scenes
from rmf_crowdsim_amd.tiles import LocalTileMesh
n = 1_000_000
pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0)
lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
for split in ("0", "1"):
    os.environ["CS_TILE_SPLIT"] = split
    mesh = LocalTileMesh(LocationHash2D(**grid), (4, 2), halo_cells=1, density_per_cell=15.0)
    scenes.add_counterflow(mesh, pts, group, scenes.CREEP_SPEED, lp, 2.0)
    for _ in range(20): mesh.step(0.05, report=False)
    [e.synchronize() for e in mesh.engines]
    t0 = time.perf_counter()
    for _ in range(100): mesh.step(0.05, report=False)
    [e.synchronize() for e in mesh.engines]
    ms = (time.perf_counter() - t0) / 100 * 1e3
    print(f"4x2 mesh of 125k-agent tiles on one GPU, split {split}: {ms:.3f} ms per mesh step = {ms/8*1e3:.1f} us per tile step")
    del mesh
