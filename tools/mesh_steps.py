#!/usr/bin/env python3
"""A walking crowd on an in-process tile mesh of the C ABI (cs_mesh_*, every tile on this GPU), stepped without reports:
what `tools/trace_mesh.sh` puts under the kernel tracer to list the launches of one tile's step.
usage: python tools/mesh_steps.py [agents] [tiles_x] [tiles_y] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmf_crowdsim_amd import LocationHash2D, Zanlungo, scenes  # noqa: E402
from rmf_crowdsim_amd.tiles import NativeTileMesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
tx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ty = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
room = scenes.WALK_SPEED * 0.05 * (steps + 30) + 4.0
pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, room=room)
mesh = NativeTileMesh(LocationHash2D(**grid), (tx, ty), 1, density_per_cell=15.0)
scenes.add_walking_crowd(mesh, pts, group, Zanlungo(*scenes.METRIC_ZANLUNGO), 2.0)
for _ in range(20):
    mesh.step(0.05, report=False)
mesh.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    mesh.step(0.05, report=False)
mesh.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{n} agents on {tx} x {ty} tiles of one GPU: {dt * 1e6:.1f} us per mesh step, {dt * 1e6 / (tx * ty):.1f} us per tile step")
