import sys, numpy as np
sys.path.insert(0, '.')
from rmf_crowdsim_amd import *
from rmf_crowdsim_amd import scenes
from rmf_crowdsim_amd.tiles import LocalTileMesh
mode = sys.argv[1]
n = 30000
pts, grid, extent, group = scenes.uniform_crowd(n, seed=5, cell_size=2.0, margin=20.0)
lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
vel = [(1.30, 0.4), (1.28, 0.4)]
single = Simulation(LocationHash2D(**grid))
mesh = LocalTileMesh(LocationHash2D(**grid), (2, 2), halo_cells=1)
for t in (single, mesh):
    for g, v in enumerate(vel):
        t.add_agents(pts[group == g], StubHighLevelPlan(v), lp, 2.0)
for k in range(61):
    if mode != 'nosingle':
        single.step(0.05, report=(mode == 'singlesync'))
    try:
        mesh.step(0.05, report=False)
    except Exception as e:
        print('step', k, 'mesh error', e)
        break
print('done', k)
