#!/usr/bin/env python3
"""Is there an agent_scale for which a 2.5 agents/m^2 head-on counter-flow at walking speed survives
1000 steps of the reference's Zanlungo model?  CPU only (the f64 oracle's arithmetic on cell-sorted
arrays, oracle_fast_steps); prints how many steps each parameter set lasts before an agent goes
non-finite or leaves the grid (what `Simulation::step` answers with Err("Index out of bounds")).

    python tests/scene_scan.py [agents] [speed]

Result: no.  The failure is geometric, not a gain instability: the larger id of a
pair never yields (weight 0, zanlungo.rs:173-198), weak forces let walkers pass through each other
(t_i = 0 -> 0/0), strong ones throw them metres per step."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_sim import fast_steps  # noqa: E402
from rmf_crowdsim_amd import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
speed = float(sys.argv[2]) if len(sys.argv) > 2 else scenes.WALK_SPEED
pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, margin=80.0)
pref = np.zeros_like(pts)
pref[group == 0, 1] = speed
pref[group == 1, 1] = -speed
order = np.concatenate([np.where(group == 0)[0], np.where(group == 1)[0]])  # ids as add_counterflow gives them
pts, pref = pts[order], pref[order]
print(f"{n} agents, 2.5 /m^2, checkerboard counter-flow at {speed} m/s, dt 0.05 s, eyesight 2 m")
for A in (10.0, 3.0, 1.0, 0.3, 0.1, 0.03, 0.01, 0.003, 0.001, 0.0001):
    for D, R in ((0.4, 0.2), (0.1, 0.2), (0.4, 0.05)):
        xy, v, lasted = pts.copy(), None, 0
        for chunk in range(200):
            xy, v, sec = fast_steps(xy, pref, (A, 1, 0, D, 2.0, R), 2.0, grid, 0.05, 5, threads=os.cpu_count(), vel=v)
            if sec < 0 or not np.isfinite(xy).all() or not np.isfinite(v).all():
                break
            lasted += 5
        print(f"  agent_scale {A:<7} force_distance {D} radius {R}: {'survives 1000 steps' if lasted >= 1000 else f'fails within {lasted + 5} steps'}", flush=True)
