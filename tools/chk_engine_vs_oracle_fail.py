#!/usr/bin/env python3
"""When a randomized source-sink scene fails on one side only: the step at which the engine and the
oracle report "Index out of bounds", and who leaves the grid.  python tools/chk_engine_vs_oracle_fail.py SEED"""
import os
import sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import test_gpu_tiles as T
from oracle_sim import OracleSimulation
from rmf_crowdsim_amd import CrowdSimError, LocationHash2D, Simulation
seed = int(sys.argv[1])
grid = dict(width=80.0, height=80.0, cell_size=float([1.0, 2.0, 2.5][seed % 3]), offset=(0.0, 0.0))
for cls in (Simulation, OracleSimulation):
    s = cls(LocationHash2D(**grid))
    plain = T._random_sink_scene(s, 900 + seed)
    prev = None
    for k in range(200 if plain else 60):
        try:
            s.step(0.1)
        except CrowdSimError as e:
            j = np.argsort(-np.maximum(prev["x"], prev["y"]))[:3]
            print(cls.__name__, "fails at step", k, e, "| furthest out before it:",
                  [(int(prev["id"][i]), float(prev["x"][i]), float(prev["y"][i]), float(prev["vx"][i]), float(prev["vy"][i])) for i in j])
            break
        prev = s.read_agents()
    else:
        print(cls.__name__, "no failure; plain", plain)
