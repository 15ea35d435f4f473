#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

The reference is Rust and cannot run in this pipeline, so these vectors are
outputs of oracle/crowdstep_oracle.cpp (f64), not of the reference itself.  They pin the oracle
against accidental change and give the GPU tests a fixture that needs no oracle build.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle_sim import OracleSimulation  # noqa: E402
from rmf_crowdsim_amd import (IdParityHighLevelPlan, LocationHash2D, MonotonicCrowd, NoLocalPlan,  # noqa: E402
                              SeededPoissonCrowd, SourceSink, StubHighLevelPlan, Zanlungo, scenes)


def agents_to_rows(arr):
    return [[int(r["id"]), float(r["x"]), float(r["y"]), float(r["vx"]), float(r["vy"]),
             int(r["next_waypoint"])] for r in arr]


def viz3():
    sim = OracleSimulation(LocationHash2D(**scenes.VIZ_GRID))
    sim.add_agents(scenes.VIZ3_POSITIONS, IdParityHighLevelPlan(scenes.VIZ_SPEED),
                   Zanlungo(*scenes.VIZ_ZANLUNGO), scenes.VIZ_EYESIGHT)
    checkpoints = {}
    for k in range(1000):
        sim.step(0.05)
        if (k + 1) in (100, 180, 200, 250, 500, 1000):
            checkpoints[str(k + 1)] = agents_to_rows(sim.read_agents())
    return {"scene": "rmf_crowdsim_viz/src/main.rs:64-94 verbatim, dt 0.05",
            "columns": ["id", "x", "y", "vx", "vy", "next_waypoint"], "checkpoints": checkpoints}


def config1():
    sim = OracleSimulation(LocationHash2D(**scenes.VIZ_GRID))
    sim.add_agents(scenes.viz_scene(256, spacing=60.0), IdParityHighLevelPlan((0.0, 0.1)),
                   Zanlungo(*scenes.VIZ_ZANLUNGO), scenes.VIZ_EYESIGHT)
    for _ in range(1000):
        sim.step(0.05)
    return {"scene": "scenes.viz_scene(256, spacing=60), id-parity +-(0,0.1), viz Zanlungo, dt 0.05, 1000 steps",
            "columns": ["id", "x", "y", "vx", "vy", "next_waypoint"],
            "final": agents_to_rows(sim.read_agents())}


def stream():
    grid = dict(width=120.0, height=120.0, cell_size=2.0, offset=(0.0, 0.0))
    sim = OracleSimulation(LocationHash2D(**grid))
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    for k in range(8):
        y = 10.0 + 2.5 * k
        left = k % 2 == 0
        sim.add_source_sink(SourceSink((10.0, y) if left else (110.0, y), 1.0, SeededPoissonCrowd(3.0, 100 + k),
                                       StubHighLevelPlan((1.3, 0.0) if left else (-1.3, 0.0)), lp,
                                       [(60.0, y) if left else (62.0, y)], False, 2.0))
    counts = []
    for _ in range(400):
        sim.step(0.05)
        counts.append([len(sim), sim.last_report["n_spawned"], sim.last_report["n_destroyed"]])
    return {"scene": "8 source-sinks, SeededPoissonCrowd(3.0, 100+k), dt 0.05, 400 steps",
            "columns": ["n_agents", "n_spawned", "n_destroyed"], "per_step": counts,
            "final": agents_to_rows(sim.read_agents())}


def dogleg(start, goal):
    """The stand-in for RMFPlanner::plan_route used by the route-follower fixture and tests."""
    import math
    mx, my = 0.5 * (start[0] + goal[0]), 0.5 * (start[1] + goal[1])
    dx, dy = goal[0] - start[0], goal[1] - start[1]
    n = math.hypot(dx, dy) or 1.0
    return [start, (mx - 2.0 * dy / n, my + 2.0 * dx / n), goal]


def route_scene(sim_cls, **kw):
    from rmf_crowdsim_amd import RouteFollower
    sim = sim_cls(LocationHash2D(160.0, 160.0, 2.0, (0.0, 0.0)), **kw)
    hlp = RouteFollower(dogleg, scale=4.0, arrive=0.1, speed=1.2)
    for k in range(8):
        y = 20.0 + 7.5 * k
        left = k % 2 == 0
        src = (20.0, y) if left else (140.0, y)
        mid = (70.0, y + 3.0) if left else (90.0, y - 3.0)
        dst = (120.0, y) if left else (40.0, y)
        sim.add_source_sink(SourceSink(src, 1.0, SeededPoissonCrowd(1.5, 40 + k), hlp, NoLocalPlan(), [mid, dst],
                                       False, 2.0))
    return sim


def route_follower():
    sim = route_scene(OracleSimulation)
    counts = []
    for _ in range(1100):
        sim.step(0.1)
        counts.append([len(sim), sim.last_report["n_spawned"], sim.last_report["n_destroyed"],
                       sim.last_report["n_waypoint_hits"]])
    return {"scene": "8 source-sinks with two waypoints each, RouteFollower(dogleg, scale 4, arrive 0.1, speed 1.2), "
                     "NoLocalPlan, dt 0.1, 1100 steps",
            "columns": ["n_agents", "n_spawned", "n_destroyed", "n_waypoint_hits"], "per_step": counts,
            "final": agents_to_rows(sim.read_agents())}


if __name__ == "__main__":
    for name, fn in (("viz3_1000_steps", viz3), ("config1_256_agents", config1), ("source_sink_stream", stream),
                     ("route_follower", route_follower)):
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(fn(), f)
        print("wrote", name)
