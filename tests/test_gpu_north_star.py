"""The north star's parity clause ("positions within 1e-4 rel of the CPU reference after 1000 steps") where it is
hardest to meet: forces of walking magnitude over the full horizon at configs[1]'s size, three ways; the headline
workload at its own 1,000,000 agents for 1000 steps; configs[4]'s 4M hotspot crowd and configs[3]'s stream at 100,000
agents against the f64 path (review of round 4: every 1000-step comparison ran where forces are 1 mm/s or exactly 0).

The f64 side of the large runs is the oracle's arithmetic on cell-sorted arrays over the host's cores
(oracle_fast_steps: the reference-shaped oracle's bits, tests/test_oracle_reference_kats.py); the f32 side of the
three-way run is the f32 build of the same code with positions kept per cell (oracle_fast_steps_ex flag 2, validated in
f64 against the plain form to 1e-10 of L: tests/test_oracle_fast_path_flags.py) and the underflow guard (flag 1), without
which an f32 reading of the reference does not survive 100 steps of a scene with real forces (asserted below).
"""
import os

import numpy as np
import pytest

from oracle_sim import OracleSimulation, fast_steps
from rmf_crowdsim_amd import (LocationHash2D, MonotonicCrowd, Simulation, SourceSink, StubHighLevelPlan, Zanlungo, _abi,
                              scenes)

pytestmark = pytest.mark.gpu
THREADS = min(16, os.cpu_count() or 1)


def crossing_flows(n, density=0.3, angle_deg=30.0, speed=scenes.WALK_SPEED, seed=5, steps=1000):
    """Two interleaved groups on a jittered lattice of `density` agents / m^2, one walking along +x, the other at
    `angle_deg` to it, both at walking speed (tests/test_gpu_parity.py::crossing_flows with room for `steps` steps)."""
    spacing = 1.0 / np.sqrt(density)
    side = int(np.ceil(np.sqrt(n)))
    pts = scenes.jittered_lattice(n, spacing, (40.0, 40.0), 0.25, seed)
    k = np.arange(n)
    group = ((k % side) + (k // side)) % 2
    th = np.radians(angle_deg)
    pref = np.where(group[:, None] == 0, np.array([speed, 0.0]), np.array([speed * np.cos(th), speed * np.sin(th)]))
    size = float(np.ceil(side * spacing + 80.0 + speed * 0.05 * steps + 40.0))
    return pts, pref, group, dict(width=size, height=size, cell_size=2.0, offset=(0.0, 0.0)), side * spacing


CROSSING_ZANLUNGO = (0.3, 1.0, 0.0, 0.4, 2.0, 0.2)


def _dist(a, xy, extent):
    return np.hypot(a["x"] - xy[:, 0], a["y"] - xy[:, 1]) / extent


def _summary(d):
    return (f"max {d.max():.2e} p99.9 {np.quantile(d, 0.999):.2e} p99 {np.quantile(d, 0.99):.2e} median {np.median(d):.2e} "
            f"beyond 1e-4: {int((d > 1e-4).sum())} ({np.mean(d > 1e-4):.1e}) beyond 1e-5: {int((d > 1e-5).sum())}")


def test_crossing_flows_at_configs1_size_for_1000_steps_three_ways():
    """Forces of walking magnitude (0.1-3 m/s, t_i of a second or two) over the north star's 1000 steps at configs[1]'s
    100,000 agents: two sparse flows crossing at 30 degrees at 1.3 m/s, the one long scene the reference's f64 path
    survives with forces of that size (tests/test_gpu_parity.py::test_crossing_flows_at_walking_speed_300_steps is its
    4,000-agent miniature).  Three legs, compared every 100 steps:
      engine (tiled == gather, bit for bit)  /  f32 build of the CPU path, per-cell positions, guarded  /  f64 CPU path.
    A dodge is a discontinuity of the MODEL (a neighbour enters the eyesight, a grazing pair's discriminant changes sign,
    t_i = min picks another pair): two readings of the same state that take such a decision one step apart end up
    centimetres, then metres, apart.  So the clause cannot hold for every agent of this scene in ANY 32-bit arithmetic;
    what the test pins is that the engine is as close to the f64 path as an independent f32 implementation of the same
    precision class, agent for agent where the scene is regular and in distribution where it is not."""
    n, steps = 100_000, 1000
    pts, pref, group, grid, extent = crossing_flows(n, steps=steps)
    lp = Zanlungo(*CROSSING_ZANLUNGO)
    th = np.radians(30.0)
    runs = {}
    for flags in (2, 1):  # tiled, gather
        sim = Simulation(LocationHash2D(**grid), flags=flags)
        ids0 = sim.add_agents(pts[group == 0], StubHighLevelPlan((scenes.WALK_SPEED, 0.0)), lp, 2.0)
        ids1 = sim.add_agents(pts[group == 1], StubHighLevelPlan((scenes.WALK_SPEED * np.cos(th), scenes.WALK_SPEED * np.sin(th))), lp, 2.0)
        runs[flags] = sim
    assert ids0[0] == 0 and ids1[-1] == n - 1
    by_id = np.concatenate([pts[group == 0], pts[group == 1]])
    pref_by_id = np.concatenate([pref[group == 0], pref[group == 1]])
    # a plain f32 reading of the reference (no guard) is struck by its own underflow flaw within the first 100 steps
    struck32 = np.zeros(n, dtype=np.uint8)
    _, _, sec32 = fast_steps(by_id, pref_by_id, CROSSING_ZANLUNGO, 2.0, grid, 0.05, 100, threads=THREADS, spurious=struck32,
                             kind="f32", cell_relative=True)
    assert sec32 < 0 or struck32.any()
    legs = {"f64": dict(kind="f64"), "f32": dict(kind="f32", guarded=True, cell_relative=True)}
    state = {k: (by_id.copy(), None) for k in legs}
    struck = np.zeros(n, dtype=np.uint8)
    strongest, dodging, cpu_s = [], 0.0, 0.0
    for chunk in range(steps // 100):
        for k, kw in legs.items():
            xy, vel = state[k]
            xy, vel, sec = fast_steps(xy, pref_by_id, CROSSING_ZANLUNGO, 2.0, grid, 0.05, 100, threads=THREADS, vel=vel,
                                      spurious=struck if k == "f64" else None, **kw)
            assert sec >= 0 and np.isfinite(xy).all() and np.isfinite(vel).all(), (k, chunk)
            state[k] = (xy, vel)
            cpu_s += sec
        assert not struck.any()  # the scene is certified on the reference's f64 path
        for sim in runs.values():
            for _ in range(99):
                sim.step(0.05, report=False)
            sim.step(0.05)
            assert sim.last_report["n_tti_zero"] == 0 and sim.last_report["n_nonfinite"] == 0
        a = runs[2].read_agents()
        assert a.tobytes() == runs[1].read_agents().tobytes()
        x64, v64 = state["f64"]
        x32 = state["f32"][0]
        force = np.hypot(v64[:, 0] - pref_by_id[:, 0], v64[:, 1] - pref_by_id[:, 1])  # |F| / m, m/s
        strongest.append(float(force.max()))
        dodging = max(dodging, float(np.mean(force > 0.01 * scenes.WALK_SPEED)))
        d_e64, d_3264 = _dist(a, x64, extent), np.hypot(*(x32 - x64).T) / extent
        d_e32 = _dist(a, x32, extent)
        print(f"crossing flows 100k, step {100 * (chunk + 1)}: max|F|/m {force.max():.2f} m/s, dodging now {np.mean(force > 0.013):.3f}\n"
              f"   engine vs f64: {_summary(d_e64)}\n   f32    vs f64: {_summary(d_3264)}\n   engine vs f32: {_summary(d_e32)}")
    print(f"CPU legs {cpu_s:.0f} s")
    assert max(strongest) > 1.0 and np.median(strongest) > 0.3 and dodging > 0.05  # forces of walking magnitude were at work
    # where the scene is regular the engine IS the f64 path to f32 resolution ...
    assert np.quantile(d_e64, 0.99) <= 1e-6 and np.median(d_e64) <= 1e-7
    # ... and nowhere is it further from it than the other f32 implementation, in distribution (the tails are the same
    # handful of chaotic dodges, not the same agents): the count beyond the north star's 1e-4, p99.9, the maximum
    beyond_e, beyond_32 = int((d_e64 > 1e-4).sum()), int((d_3264 > 1e-4).sum())
    assert beyond_e <= 2 * beyond_32 + 10
    assert np.quantile(d_e64, 0.999) <= 3.0 * max(np.quantile(d_3264, 0.999), 1e-6)
    assert d_e64.max() <= 10.0 * d_3264.max()
    # engine vs the f32 leg: the same, agent for agent, wherever neither has taken a dodge differently
    assert np.quantile(d_e32, 0.99) <= 1e-6
    # the clause itself, on the agents the model lets it hold for: all but a few in 10,000
    assert np.mean(d_e64 > 1e-4) <= 5e-4


def _headline_scene(n, steps, walking):
    import bench
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, room=bench.walk_room(steps) if walking else 0.0)
    speed = min(scenes.CREEP_SPEED, 0.25 / (steps + 2))  # bench.py's rule: at most 2.5 cm of closing over the run
    return pts, grid, extent, group, speed


@pytest.mark.parametrize("workload", ["walk", "creep"])
def test_the_headline_workload_at_one_million_agents_for_1000_steps(workload):
    """BASELINE.json's metric and its parity clause in ONE run: 1,000,000 agents (bench.py's default scene, walking, and
    its standing twin with the non-zero forces), dt 0.05 s, 1000 steps, positions within 1e-4 of the extent of the f64
    CPU path.  The reference's f64 path loses about one agent per 3e6 agent-steps to its underflow flaw (DESIGN.md
    section 5); those, named by the CPU side itself, are left out and nobody else; the engine stays finite."""
    n, steps = 1_000_000, 1000
    walking = workload == "walk"
    pts, grid, extent, group, speed = _headline_scene(n, steps, walking)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    sim = Simulation(LocationHash2D(**grid), capacity_hint=n + 1024)
    if walking:
        ids = np.asarray(scenes.add_walking_crowd(sim, pts, group, lp, 2.0, creep=speed))
    else:
        ids = np.asarray(scenes.add_counterflow(sim, pts, group, speed, lp, 2.0))
    for k in range(steps - 1):
        sim.step(0.05, report=False)
    sim.step(0.05)
    rep = sim.last_report
    assert rep["n_tti_zero"] == 0 and rep["n_nonfinite"] == 0 and rep["n_agents"] == n
    a = sim.read_agents()
    by_id = np.empty_like(pts)
    by_id[ids] = pts
    pref = np.zeros_like(pts)
    pref[ids, 1] = np.where(group == 0, speed, -speed)
    if walking:
        pref[:, 0] = scenes.WALK_SPEED
    struck = np.zeros(n, dtype=np.uint8)
    xy, vel, sec = fast_steps(by_id, pref, scenes.METRIC_ZANLUNGO, 2.0, grid, 0.05, steps, threads=THREADS, spurious=struck)
    assert sec > 0
    ok = np.isfinite(xy).all(axis=1)
    assert np.isfinite(a["x"]).all() and np.isfinite(a["vx"]).all()
    assert ((~ok) == (struck != 0)).all() and (~ok).sum() <= 2000  # (661 when this was written: dense cells meet the flaw more often)0
    d = _dist(a, xy, extent)[ok]
    force = np.hypot(vel[ok, 0] - pref[ok, 0], vel[ok, 1] - pref[ok, 1])
    print(f"headline scene ({workload}), 1M agents x 1000 steps: engine vs f64: {_summary(d)}; {int((~ok).sum())} agents NaN on "
          f"the reference's f64 path; forced {float(np.mean(force > 0)):.3f}; CPU side {sec:.0f} s")
    assert d.max() <= 1e-4
    if walking:
        assert abs(float((a["x"] - by_id[:, 0]).mean()) - steps * 0.05 * scenes.WALK_SPEED) < 1e-2
    else:
        assert np.mean(force > 0) > 0.9


def test_the_scattered_crowd_at_configs1_size_for_1000_steps_three_ways():
    """The third scene bench.py times (`scattered_scene`: a randomly thinned lattice, neighbour counts that scatter like a
    real crowd's, forces of every size the creep leaves) at configs[1]'s 100,000 agents for the north star's 1000 steps:
    engine / f32 leg of the CPU path / f64 CPU path.  The reference's f64 path is struck by its own underflow flaw far
    more often here than on the lattice (neighbours at every distance: ~760 agents of 100,000); they are named by the
    CPU side and left out, nobody else."""
    n, steps = 100_000, 1000
    pts, grid, extent, group = scenes.random_crowd(n, seed=7, cell_size=2.0)
    speed = min(scenes.CREEP_SPEED, 0.25 / (steps + 2))
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    sim = Simulation(LocationHash2D(**grid))
    ids = np.asarray(scenes.add_counterflow(sim, pts, group, speed, lp, 2.0))
    for k in range(steps - 1):
        sim.step(0.05, report=False)
    sim.step(0.05)
    assert sim.last_report["n_tti_zero"] == 0 and sim.last_report["n_nonfinite"] == 0 and sim.last_report["n_agents"] == n
    a = sim.read_agents()
    by_id = np.empty_like(pts)
    by_id[ids] = pts
    pref = np.zeros_like(pts)
    pref[ids, 1] = np.where(group == 0, speed, -speed)
    struck = np.zeros(n, dtype=np.uint8)
    x64, v64, sec = fast_steps(by_id, pref, scenes.METRIC_ZANLUNGO, 2.0, grid, 0.05, steps, threads=THREADS, spurious=struck)
    x32, _, sec32 = fast_steps(by_id, pref, scenes.METRIC_ZANLUNGO, 2.0, grid, 0.05, steps, threads=THREADS, kind="f32",
                               guarded=True, cell_relative=True)
    assert sec > 0 and sec32 > 0 and np.isfinite(x32).all() and np.isfinite(a["x"]).all()
    ok = np.isfinite(x64).all(axis=1)
    assert ((~ok) == (struck != 0)).all() and (~ok).sum() <= 3000
    d_e64 = _dist(a, x64, extent)[ok]
    d_3264 = (np.hypot(*(x32 - x64).T) / extent)[ok]
    d_e32 = _dist(a, x32, extent)
    force = np.hypot(v64[ok, 0] - pref[ok, 0], v64[ok, 1] - pref[ok, 1])
    print(f"scattered crowd 100k x 1000 steps: {int((~ok).sum())} agents NaN on the reference's f64 path; forced "
          f"{float(np.mean(force > 0)):.3f}\n   engine vs f64: {_summary(d_e64)}\n   f32    vs f64: {_summary(d_3264)}\n"
          f"   engine vs f32: {_summary(d_e32)}")
    assert d_e64.max() <= 1e-4 and np.mean(force > 0) > 0.7
    assert d_e64.max() <= 10.0 * max(d_3264.max(), 1e-7)  # the same order as the other f32 implementation


def test_config4_four_million_hotspot_agents_against_the_f64_path():
    """configs[4] at full size against the f64 CPU path (round 4 checked it through properties only): 4M agents, half of
    them in Gaussian hotspots of up to 4.9 agents / m^2 (neighbour lists beyond 64 entries: CS_CFG_DENSE, windows walked
    in chunks), 6 steps, every agent compared."""
    n, steps = 4_000_000, 6
    pts, grid, extent, group = scenes.hotspot_crowd(n, seed=7, cell_size=2.0)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    sim = Simulation(LocationHash2D(**grid), flags=2 | _abi.CS_CFG_DENSE, capacity_hint=n + 1024)
    ids = np.asarray(scenes.add_counterflow(sim, pts, group, scenes.CREEP_SPEED, lp, 2.0))
    for k in range(steps - 1):
        sim.step(0.05, report=False)
    sim.step(0.05)
    rep = sim.last_report
    assert rep["n_tti_zero"] == 0 and rep["n_nonfinite"] == 0 and rep["n_agents"] == n
    a = sim.read_agents()
    by_id = np.empty_like(pts)
    by_id[ids] = pts
    pref = np.zeros_like(pts)
    pref[ids, 1] = np.where(group == 0, scenes.CREEP_SPEED, -scenes.CREEP_SPEED)
    struck = np.zeros(n, dtype=np.uint8)
    xy, vel, sec = fast_steps(by_id, pref, scenes.METRIC_ZANLUNGO, 2.0, grid, 0.05, steps, threads=THREADS, spurious=struck)
    assert sec > 0
    ok = np.isfinite(xy).all(axis=1)
    assert ((~ok) == (struck != 0)).all() and (~ok).sum() <= 2000  # (661 when this was written: dense cells meet the flaw more often)
    d = _dist(a, xy, extent)[ok]
    force = np.hypot(vel[ok, 0], np.abs(vel[ok, 1]) - scenes.CREEP_SPEED)
    dv = np.hypot(a["vx"][ok] - vel[ok, 0], a["vy"][ok] - vel[ok, 1])
    print(f"configs[4], 4M agents x {steps} steps: engine vs f64: {_summary(d)}; |dv| p99.9 / max|F| "
          f"{np.quantile(dv, 0.999) / force.max():.2e}; forced {float(np.mean(force > 0)):.3f}; {int((~ok).sum())} NaN on the "
          f"reference's path; CPU side {sec:.0f} s")
    assert d.max() <= 1e-4 and np.mean(force > 0) > 0.75  # (0.83: the thin background between the hotspots sees fewer neighbours)
    assert np.quantile(dv, 0.999) <= 2e-3 * force.max()


def test_config3_stream_of_one_hundred_thousand_agents_against_the_oracle():
    """configs[3] at a tenth of its size against the reference-shaped oracle (round 4: 10k agents): 2,500 source-sink
    lanes sustaining ~100,000 walkers, every step's spawn and destroy counts, and ids / waypoint counters / positions at
    the end of 200 steps past the fill (~75 releases and as many arrivals per lane)."""
    lanes, grid, fill_steps = scenes.stream_lanes(100_000, cell_size=2.0)
    assert len(lanes) == 2_500
    steps = fill_steps + 200
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    out = {}
    for name, cls in (("engine", Simulation), ("oracle", OracleSimulation)):
        sim = cls(LocationHash2D(**grid))
        plans = {}
        for src, dst, vel in lanes:
            hlp = plans.setdefault(vel, StubHighLevelPlan(vel))
            sim.add_source_sink(SourceSink(src, 0.5, MonotonicCrowd(1000.0), hlp, lp, [dst], False, 2.0))
        counts = []
        if name == "oracle":
            sim.degenerate_flips()
        for k in range(steps):
            sim.step(0.05)
            r = sim.last_report
            counts.append((r["n_spawned"], r["n_destroyed"], r["n_agents"], r["n_tti_zero"], r["n_nonfinite"]))
        out[name] = (sim.read_agents(), np.array(counts), sim.degenerate_flips() if name == "oracle" else 0)
    (a, ca, _), (b, cb, flips) = out["engine"], out["oracle"]
    assert (ca == cb).all() and ca[:, 3:].sum() == 0
    past_fill = ca[fill_steps:]
    assert len(a) > 80_000 and past_fill[:, 0].sum() > 60_000 and past_fill[:, 1].sum() > 60_000
    assert (a["id"] == b["id"]).all() and (a["next_waypoint"] == b["next_waypoint"]).all()
    extent = grid["width"]
    d = np.hypot(a["x"] - b["x"], a["y"] - b["y"]) / extent
    dv = np.hypot(a["vx"] - b["vx"], a["vy"] - b["vy"]) / scenes.WALK_SPEED
    print(f"configs[3] at 100k: {len(a)} alive after {steps} steps, {int(ca[:, 0].sum())} spawned, {int(ca[:, 1].sum())} destroyed; "
          f"engine vs oracle: {_summary(d)}; |dv|/v max {dv.max():.2e}; degenerate flips on the f64 path: {flips}")
    assert d.max() <= 1e-4
