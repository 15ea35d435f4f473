import os, sys, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("CS_TILE_SPLIT", "0")
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
        for i, e in enumerate(mesh.engines):
            try:
                e.synchronize()
            except CrowdSimError as err:
                print("tile", i, "raised:", err)
        # the state before the failing step
        for i, a in enumerate(prev):
            r = mesh.layout.rect(*mesh.layout.coords(i))
            cx = np.floor(a["x"] / cell).astype(int); cy = np.floor(a["y"] / cell).astype(int)
            dx = np.minimum(cx - r[0], r[1] - 1 - cx); dy = np.minimum(cy - r[2], r[3] - 1 - cy)
            print("tile", i, "rect", r, "n", len(a), "min dist to edge x", dx.min() if len(a) else None, "y", dy.min() if len(a) else None)
            # distance to the tile's LOCAL grid boundary (rect + ring, clipped to the domain)
            n_cells = int(grid["width"] / cell)
            l0, l1 = max(r[0] - halo, 0), min(r[1] + halo, n_cells); m0, m1 = max(r[2] - halo, 0), min(r[3] + halo, n_cells)
            ex = np.minimum(a["x"] / cell - l0, l1 - a["x"] / cell); ey = np.minimum(a["y"] / cell - m0, m1 - a["y"] / cell)
            order = np.argsort(np.minimum(ex, ey))[:6]
            print("  closest to the local grid's boundary:", [(int(a["id"][j]), round(float(a["x"][j]), 3), round(float(a["y"][j]), 3), round(float(a["vx"][j]), 3), round(float(a["vy"][j]), 3)) for j in order])
        break
    prev = [e.read_agents() for e in mesh.engines]
else:
    print("no error")
