#!/usr/bin/env python3
"""Where the wave cycles of the tiled step kernel go.  Rebuilds the engine with -DCS_PHASE_CLOCKS
(profiling build: every wave adds its cycle count per phase to a device array), runs the bench
scene and prints the shares.  Run on the GPU box; the product library is rebuilt afterwards.

    python tools/phase_clocks.py [--agents N] [--eyesight E] [--cell C] [--steps K]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--agents", type=int, default=1_000_000)
ap.add_argument("--eyesight", type=float, default=2.0)
ap.add_argument("--cell", type=float, default=2.0)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--workload", choices=["uniform", "hotspots", "random", "walk"], default="uniform")
args = ap.parse_args()

os.environ["CS_HIPCC_EXTRA"] = "-DCS_PHASE_CLOCKS"
from rmf_crowdsim_amd import _native  # noqa: E402

_native.build(force=True)
try:
    import numpy as np  # noqa: E402
    from rmf_crowdsim_amd import scenes  # noqa: E402
    from rmf_crowdsim_amd.simulation import IdParityHighLevelPlan, Simulation, Zanlungo  # noqa: E402

    crowd = {"hotspots": scenes.hotspot_crowd, "random": scenes.random_crowd}.get(args.workload, scenes.uniform_crowd)
    if args.workload == "walk":  # bench.py's default: the crowd walks +x, with room on that side
        room = scenes.WALK_SPEED * 0.05 * (args.steps + 18) + 4.0
        pts, grid, extent, group = scenes.uniform_crowd(args.agents, seed=7, cell_size=args.cell, room=room)
    else:
        pts, grid, extent, group = crowd(args.agents, seed=7, cell_size=args.cell)
    from rmf_crowdsim_amd.simulation import LocationHash2D
    sim = Simulation(LocationHash2D(**grid), flags=4 if args.workload == "hotspots" else 0)  # 4 = CS_CFG_DENSE
    if args.workload == "walk":
        scenes.add_walking_crowd(sim, pts, group, Zanlungo(*scenes.METRIC_ZANLUNGO), args.eyesight)
    else:
        scenes.add_counterflow(sim, pts, group, scenes.CREEP_SPEED, Zanlungo(*scenes.METRIC_ZANLUNGO), args.eyesight)
    for _ in range(10):
        sim.step(0.05, report=False)
    lib = sim._lib
    fn = lib.cs_debug_phase_cycles
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    fn.restype = None
    out = (C.c_ulonglong * 12)()
    fn(sim._engine, out, 1)
    for _ in range(args.steps):
        sim.step(0.05, report=False)
    fn(sim._engine, out, 1)
    names = ["descriptor", "geometry", "staging: loads", "staging: barrier 1", "staging: placement",
             "staging: barrier 2", "agent setup", "filter", "time-to-collision", "forces", "epilogue"]
    total = float(sum(out[:11])) or 1.0
    print(f"agents {args.agents} eyesight {args.eyesight} cell {args.cell}: wave cycles per step {total / args.steps:.4g}")
    print(f"  windows per step off the LDS path: {(out[11] & 0xFFFFFFFF) / 4.0 / args.steps:.2f}, walked in chunks: {(out[11] >> 32) / 4.0 / args.steps:.2f}")
    for n, v in zip(names, out):
        print(f"  {n:18s} {100.0 * v / total:5.1f} %   {v / args.steps / (args.agents / 64.0):8.0f} cycles per wave of 64 agents")
finally:
    del os.environ["CS_HIPCC_EXTRA"]
    _native.build(force=True)
