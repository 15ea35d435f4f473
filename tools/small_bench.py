import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from rmf_crowdsim_amd import LocationHash2D, Simulation, Zanlungo, scenes
lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
for n in (4000, 125000, 250000):
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0)
    sim = Simulation(LocationHash2D(**grid))
    scenes.add_counterflow(sim, pts, group, 1e-5, lp, 2.0)
    for _ in range(50): sim.step(0.05, report=False)
    sim.synchronize()
    for steps in (200, 2000):
        t0 = time.perf_counter()
        for _ in range(steps): sim.step(0.05, report=False)
        t1 = time.perf_counter()
        sim.synchronize()
        t2 = time.perf_counter()
        print(f"n {n} steps {steps}: issue {1e6*(t1-t0)/steps:.1f} us/step, total {1e6*(t2-t0)/steps:.1f} us/step", flush=True)
