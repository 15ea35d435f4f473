import os, sys, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ["CS_TILE_SPLIT"] = "1"
import numpy as np
import test_gpu_tiles as T
from rmf_crowdsim_amd import CrowdSimError, LocationHash2D, Simulation
from rmf_crowdsim_amd.tiles import LocalTileMesh
seed = int(sys.argv[1])
cell = float([1.0, 2.0, 2.5][seed % 3])
grid = dict(width=80.0, height=80.0, cell_size=cell, offset=(0.0, 0.0))
tiles = [(2, 2), (3, 1), (1, 2)][seed % 3]
halo = math.ceil(3.0 / cell)
mesh = LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=halo)
plain = T._random_sink_scene(mesh, 900 + seed)
print("cell", cell, "tiles", tiles, "halo", halo, "plain", plain, [mesh.layout.rect(*mesh.layout.coords(i)) for i in range(mesh.layout.n_tiles)])
prev = None
for k in range(200):
    try:
        mesh.step(0.1)
    except CrowdSimError as e:
        print("step", k, "error", e)
        # the state before the failing step
        for i, a in enumerate(prev):
            r = mesh.layout.rect(*mesh.layout.coords(i))
            cx = np.floor(a["x"] / cell).astype(int); cy = np.floor(a["y"] / cell).astype(int)
            dx = np.minimum(cx - r[0], r[1] - 1 - cx); dy = np.minimum(cy - r[2], r[3] - 1 - cy)
            print("tile", i, "rect", r, "n", len(a), "min dist to edge x", dx.min() if len(a) else None, "y", dy.min() if len(a) else None)
            near = (dx <= halo + 1) | (dy <= halo + 1)
            print("  agents within halo+1 of an edge:", a[near][["id", "x", "y", "vx", "vy"]][:12])
        break
    prev = [e.read_agents() for e in mesh.engines]
else:
    print("no error")
