#!/usr/bin/env python3
"""Looks for performance cliffs: the bench crowd at other densities, cell sizes and eyesight
ranges (1M agents).  Run on the GPU box:  python tools/config_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
from rmf_crowdsim_amd import LocationHash2D, Simulation, Zanlungo, _abi, scenes  # noqa: E402


def run(n, density, cell, eyesight, flags=0):
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, density=density, cell_size=cell)
    sim = Simulation(LocationHash2D(**grid), flags=flags)
    scenes.add_counterflow(sim, pts, group, 1e-4, Zanlungo(*scenes.METRIC_ZANLUNGO), eyesight)
    for _ in range(5):
        sim.step(0.05, report=False)
    sim.synchronize()
    sim.profile_reset()
    sim.profile_enable(1 << _abi.CS_K_NEIGHBOUR_FORCE)
    t0 = time.perf_counter()
    for _ in range(20):
        sim.step(0.05, report=False)
    sim.synchronize()
    el = (time.perf_counter() - t0) / 20
    sim.profile_enable(0)
    k4 = sim.profile_read()["neighbour_force"]
    sim.step(0.05)
    r = sim.last_report
    print(f"density {density:5.2f} cell {cell:4.1f} eyesight {eyesight:4.1f}: step {el * 1e3:8.3f} ms  "
          f"K4 {k4['total_ms'] / max(k4['launches'], 1):8.3f} ms  {n / el:9.3g} agent-steps/s  "
          f"tti0 {r['n_tti_zero']} nonfinite {r['n_nonfinite']}", flush=True)


if __name__ == "__main__":
    n = 1_000_000
    if len(sys.argv) >= 4:  # one configuration: density cell eyesight [engine flags]
        run(n, float(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
        sys.exit(0)
    for density in (0.1, 0.5, 1.0, 2.5, 4.0):
        run(n, density, 2.0, 2.0)
    for cell in (0.5, 1.0, 4.0, 8.0):
        run(n, 2.5, cell, 2.0)
    for eyesight in (1.0, 4.0, 6.0):
        run(n, 2.5, 2.0, eyesight)
    run(n, 0.5, 2.0, 6.0)
    run(n, 0.1, 0.5, 2.0)
