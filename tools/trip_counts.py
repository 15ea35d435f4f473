#!/usr/bin/env python3
"""Per-wave loop trips of the tiled step kernel on a bench workload (diagnostic build, -DCS_TILE_TRIPS).

usage (under gpurun, after `bash tools/build_variant.sh trips -DCS_TILE_TRIPS`):
    CS_LIB_PATH=$PWD/rmf_crowdsim_amd/lib/variants/trips.so python tools/trip_counts.py [walk|creep|random|hotspots] [agents]
Prints, per wave and step: filter trips (two candidates each), time-to-collision trips (two entries each), force
trips, and the shares of waves that went beyond their LDS rows / had to drain their lists.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rmf_crowdsim_amd import LocationHash2D, Simulation, Zanlungo, scenes  # noqa: E402
from rmf_crowdsim_amd import _abi  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "walk"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    steps = 20
    if workload in ("walk", "creep"):
        sim, _, _ = bench.build_crowd(Simulation, n, 2.0, 2.0, scenes.CREEP_SPEED, workload=workload, steps=steps)
    else:
        crowd = {"hotspots": scenes.hotspot_crowd, "random": scenes.random_crowd}[workload]
        pts, grid, _, group = crowd(n, seed=7, cell_size=2.0)
        flags = _abi.CS_CFG_DEFAULT | (_abi.CS_CFG_DENSE if workload == "hotspots" else 0)
        sim = Simulation(LocationHash2D(**grid), flags=flags)
        bench.populate(sim, workload, pts, group, scenes.CREEP_SPEED, Zanlungo(*scenes.METRIC_ZANLUNGO), 2.0)
    L = sim._lib
    for _ in range(steps):
        sim.step(0.05)
    sim.synchronize()
    d = [L.cs_kernel_stat(sim._engine, 100 + i) for i in range(12)]
    waves = max(d[0], 1)
    print(f"{workload} {n} agents, {steps} steps: waves/step {d[0] / steps:.0f}  beyond LDS rows {d[1] / waves:.3f}  "
          f"drained {d[2] / waves:.3f}")
    print(f"  per wave: filter trips {d[5] / waves:.1f}  ttc trips {d[4] / waves:.1f} (entries per lane {d[7] / waves / 64:.1f})  "
          f"force trips {d[3] / waves:.1f} (entries per lane {d[6] / waves / 64:.1f})")
    print(f"  time-to-collision passes repeated in the guarded form: {d[8] / waves:.3f} per wave")
    agents, finite = max(d[9], 1), d[10]
    print(f"  agents on the LDS path per step {d[9] / steps:.0f}, with a finite time to collision {finite / agents:.4f}")
    # the summary bench.py quotes per scene (bench.scene_stats; merged into profiles/rNN/scene_stats.json)
    import json

    class A:
        cell, eyesight = 2.0, 2.0
    key = bench.profile_key(f"scene {workload} agents {n} cell {A.cell} eyesight {A.eyesight}")
    stats = {
        "steps_counted": steps,
        "mean_neighbours": d[7] / agents - 1.0,                 # listed entries per agent, the agent itself taken off
        "finite_ttc_frac": finite / agents,                       # agents whose force pass runs (t_i != +inf)
        "mean_forward_neighbours": d[6] / agents,                 # entries with the larger id (non-zero weight), per agent
        "mean_forward_neighbours_of_finite": d[6] / max(finite, 1),
        "filter_trips_per_wave": d[5] / waves, "ttc_trips_per_wave": d[4] / waves, "force_trips_per_wave": d[3] / waves,
        "force_lane_use": (d[6] / waves / 64.0) / max(d[3] / waves, 1e-9),
        "waves_beyond_lds_rows_frac": d[1] / waves,
    }
    print("SCENE_STATS " + json.dumps({key: {"workload": workload, "agents": n, **stats}}))


if __name__ == "__main__":
    main()
