"""BASELINE.json configs[1], [3] and [4] at the sizes they are written with, on the device.

configs[1] (100k agents on 200 x 200 m) is small enough for the f64 oracle (0.25 s per step), so it
is compared with it directly, plus the three-way f32 / f64 split of SURVEY.md section 8d.  configs[3]
(1M agents sustained by source-sinks) and configs[4] (4M agents, half of them in hotspots) are far
beyond what the oracle steps in seconds: they are checked through properties that need no oracle run
(the exact gather kernel, the LDS-tiled kernel and a tile mesh must give the same bits; nobody is
lost, duplicated or non-finite; spawn and destroy counts add up).  configs[0] and [2] at their sizes:
tests/test_gpu_parity.py::test_config1_256_agents_1000_steps,
tests/test_gpu_tiles.py::test_full_size_crowd_invariants.
"""
import numpy as np
import pytest

from oracle_sim import OracleSimulation, OracleSimulationF32
from rmf_crowdsim_amd import (LocationHash2D, MonotonicCrowd, Simulation, SourceSink, StubHighLevelPlan,
                              Zanlungo, _abi, scenes)
from rmf_crowdsim_amd.tiles import LocalTileMesh

pytestmark = pytest.mark.gpu


def _rel(a, b, scale, ok=None):
    assert (a["id"] == b["id"]).all()
    dp = np.hypot(a["x"] - b["x"], a["y"] - b["y"])
    return float((dp if ok is None else dp[ok]).max() / scale)


def test_config1_exact_size():
    """configs[1] as written: 100,000 agents uniform on 200 x 200 m (2.5 / m^2), Zanlungo, cell 2 m,
    eyesight 2 m, dt 0.05 s, one MI355X.  Engine vs f64 oracle over 8 steps (every step compared),
    then the three-way split: engine (cell-relative f32) / oracle built in f32 / f64 oracle.

    The reference's f64 path has an underflow flaw of its own: a pair whose relative
    velocity is ~1e-162 reads as "colliding now" (t_i = 0) and the larger id goes NaN; it strikes 5
    of these 100,000 agents in steps 3-4.  f32 flushes such forces to exactly 0, so the engine
    cannot hit it: it must stay finite with n_tti_zero = 0, and those agents are left out."""
    n = 100_000
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0)
    assert abs(extent - 200.0) < 1.0  # 317 x 317 lattice sites, 0.632 m apart
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    sims = {}
    for name, cls in (("gpu", Simulation), ("o32", OracleSimulationF32), ("o64", OracleSimulation)):
        sims[name] = cls(LocationHash2D(**grid))
        scenes.add_counterflow(sims[name], pts, group, scenes.CREEP_SPEED, lp, 2.0)
    sims["o64"].spurious_victims()  # (starts the oracle's record of the agents its f64 flaw strikes)
    worst = 0.0
    for k in range(8):
        for s in sims.values():
            s.step(0.05)
        a, b = sims["gpu"].read_agents(), sims["o64"].read_agents()
        assert len(a) == n and np.isfinite(a["x"]).all() and np.isfinite(a["vx"]).all()
        ok = np.isfinite(b["x"])
        # left out: exactly the agents the oracle names as victims of the flaw, nobody else (5 when this was written)
        assert set(int(i) for i in b["id"][~ok]) == sims["o64"].spurious_victims() and (~ok).sum() <= 10
        worst = max(worst, _rel(a, b, extent, ok))
        assert sims["gpu"].last_report["n_tti_zero"] == 0 and sims["gpu"].last_report["n_nonfinite"] == 0
    c = sims["o32"].read_agents()
    ok = ok & np.isfinite(c["x"])
    e_gpu_64, e_32_64, e_gpu_32 = _rel(a, b, extent, ok), _rel(c, b, extent, ok), _rel(a, c, extent, ok)
    force = np.hypot(b["vx"], np.abs(b["vy"]) - scenes.CREEP_SPEED)[ok]
    dv = np.hypot(a["vx"] - b["vx"], a["vy"] - b["vy"])[ok]
    print(f"configs[1] 100k: worst |dp|/L over 8 steps {worst:.2e}; gpu-f64 {e_gpu_64:.2e}, f32-f64 {e_32_64:.2e}, "
          f"gpu-f32 {e_gpu_32:.2e}; forced {float(np.mean(force > 0)):.3f}; |dv| p99.9/max|F| "
          f"{float(np.quantile(dv, 0.999) / force.max()):.2e}; reference-path NaN agents {int((~ok).sum())}")
    assert worst <= 1e-4 and np.mean(force > 0) > 0.95
    assert np.quantile(dv, 0.999) <= 2e-3 * force.max()
    # the engine's cell-relative f32 must not be further from the f64 path than plain f32 is
    assert e_gpu_64 <= max(e_32_64, 1e-7) * 1.5


def test_config1_full_size_for_the_north_star_s_1000_steps():
    """The north star's parity clause at configs[1]'s size: 100,000 agents on 200 x 200 m, dt 0.05 s,
    1000 steps, positions within 1e-4 (relative to the extent) of the CPU reference path.  The f64
    side is the oracle's arithmetic on cell-sorted arrays spread over the host's cores
    (oracle_fast_steps: bit-identical to the reference-shaped oracle, tests/test_oracle_reference_kats.py;
    the reference-shaped one would take four minutes).  The reference's f64 path loses about one agent
    per 1e6 agent-steps to its underflow flaw: those are left out, the engine must
    stay finite."""
    import os
    from oracle_sim import fast_steps
    n, steps = 100_000, 1000
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    sim = Simulation(LocationHash2D(**grid))
    ids = scenes.add_counterflow(sim, pts, group, scenes.CREEP_SPEED, lp, 2.0)
    for k in range(steps - 1):
        sim.step(0.05, report=False)
    sim.step(0.05)
    assert sim.last_report["n_tti_zero"] == 0 and sim.last_report["n_nonfinite"] == 0
    a = sim.read_agents()
    # the same crowd for the CPU side, in id order (ids follow add_counterflow: group 0 first)
    by_id = np.empty_like(pts)
    by_id[ids] = pts
    pref = np.zeros_like(pts)
    pref[ids, 1] = np.where(group == 0, scenes.CREEP_SPEED, -scenes.CREEP_SPEED)
    struck = np.zeros(n, dtype=np.uint8)
    xy, vel, sec = fast_steps(by_id, pref, scenes.METRIC_ZANLUNGO, 2.0, grid, 0.05, steps,
                              threads=min(16, os.cpu_count() or 1), spurious=struck)
    assert sec > 0
    ok = np.isfinite(xy).all(axis=1)
    # left out: exactly the agents struck by the flaw (the CPU side reports them), nobody else (30 when this was written)
    assert np.isfinite(a["x"]).all() and np.isfinite(a["vx"]).all()
    assert ((~ok) == (struck != 0)).all() and (~ok).sum() <= 60
    dp = np.hypot(a["x"] - xy[:, 0], a["y"] - xy[:, 1])[ok]
    force = np.hypot(vel[ok, 0], np.abs(vel[ok, 1]) - scenes.CREEP_SPEED)
    print(f"configs[1], 1000 steps: |dp|/L = {dp.max() / extent:.2e} (p99.9 {np.quantile(dp, 0.999) / extent:.2e}); "
          f"{int((~ok).sum())} agents NaN on the reference's f64 path; forced {float(np.mean(force > 0)):.3f}; "
          f"CPU side {sec:.1f} s")
    assert dp.max() / extent <= 1e-4 and np.mean(force > 0) > 0.9


def _stream(target, lanes, lp, eyesight):
    plans = {}
    for src, dst, vel in lanes:
        hlp = plans.setdefault(vel, StubHighLevelPlan(vel))
        target.add_source_sink(SourceSink(src, 0.5, MonotonicCrowd(1000.0), hlp, lp, [dst], False, eyesight))


def test_config3_one_million_agents_sustained_by_source_sinks():
    """configs[3] at full size: 25,000 source-sink lanes (16 m each, one agent released whenever the
    source is free: lib.rs:212-217) fill up to ~1M walking agents (1.3 m/s); from then on every step
    spawns and destroys thousands (per-step compaction).  The exact gather kernel, the tiled kernel
    and a 1 x 1 tile mesh (the multi-GPU bookkeeping: device-side spawn flags, halo buffers) must
    agree bit for bit; ids are unique, the population is conserved (alive = spawned - destroyed),
    everyone alive is finite and inside its lane."""
    lanes, grid, fill_steps = scenes.stream_lanes(1_000_000, cell_size=2.0)
    assert len(lanes) == 25_000
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    steps = fill_steps + 40
    runs = {}
    for name in ("tiled", "gather", "mesh"):
        if name == "mesh":
            t = LocalTileMesh(LocationHash2D(**grid), (1, 1), halo_cells=1, capacity_hint=1_200_000)
        else:
            t = Simulation(LocationHash2D(**grid), flags=2 if name == "tiled" else 1, capacity_hint=1_200_000)
        _stream(t, lanes, lp, 2.0)
        spawned = destroyed = late_spawned = late_destroyed = 0
        for k in range(steps):
            # the two engines report every step (host path: counts add up exactly); the mesh runs
            # fire-and-forget (spawn flags and ids stay on the device)
            t.step(0.05, report=name != "mesh")
            if name != "mesh":
                spawned += t.last_report["n_spawned"]
                destroyed += t.last_report["n_destroyed"]
                if k >= steps - 28:  # the lanes release in lock-step, every 7th step (0.4 m at 1.3 m/s)
                    late_spawned += t.last_report["n_spawned"]
                    late_destroyed += t.last_report["n_destroyed"]
        runs[name] = (t.read_agents(), spawned, destroyed, t, late_spawned, late_destroyed)
    a, spawned, destroyed, sim, late_spawned, late_destroyed = runs["tiled"]
    g, g_spawned, g_destroyed = runs["gather"][:3]
    m = runs["mesh"][0]
    print(f"configs[3]: {len(a)} agents alive after {steps} steps, {spawned} spawned, {destroyed} destroyed, "
          f"last 28 steps +{late_spawned} -{late_destroyed}")
    assert 800_000 < len(a) <= 1_050_000
    assert late_spawned >= 75_000 and late_destroyed >= 75_000  # steady state: births balance deaths
    assert len(np.unique(a["id"])) == len(a)
    assert a["id"].max() < spawned + 1 and spawned - destroyed == len(a)
    assert (spawned, destroyed) == (g_spawned, g_destroyed)
    assert a.tobytes() == g.tobytes() == m.tobytes()
    assert np.isfinite(a["x"]).all() and np.isfinite(a["vx"]).all()
    assert sim.last_report["n_tti_zero"] == 0 and sim.last_report["n_nonfinite"] == 0
    # walkers stay in their lanes (lanes are 1 m apart; the force is perpendicular but small here)
    lane_x = np.array([l[0][0] for l in lanes])
    nearest = np.abs(a["x"][:, None][:2000] - lane_x[None, :]).min(axis=1)
    assert nearest.max() < 0.5


def test_two_hundred_thousand_source_sinks_on_one_engine():
    """Every source-sink is a planner group, and the group index travels in the agents' `meta` word: 16 bits of
    it until an engine has more than 65,535 groups, 20 from then on (the live agents' words are re-packed in
    place; round 2 refused the 65,536th sink and bench.py lengthened its lanes to stay below).  configs[3]'s
    stream cut into 200,000 short lanes (2 m, five walkers each): the switch happens in the middle of
    registration AFTER a first batch of sinks has already spawned agents, so words of both packings are
    re-packed.  Tiled = gather = 1 x 1 mesh bit for bit; ids unique; alive = spawned - destroyed."""
    lanes, grid, fill_steps = scenes.stream_lanes(1_000_000, lane_length=2.0, cell_size=2.0)
    assert len(lanes) == 200_000
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    runs = {}
    for name in ("tiled", "gather", "mesh"):
        if name == "mesh":
            t = LocalTileMesh(LocationHash2D(**grid), (1, 1), halo_cells=1, capacity_hint=1_200_000)
        else:
            t = Simulation(LocationHash2D(**grid), flags=2 if name == "tiled" else 1, capacity_hint=1_200_000)
        _stream(t, lanes[:60_000], lp, 2.0)
        for _ in range(12):  # agents of the first sinks exist (16-bit packing) when the 65,536th group arrives
            t.step(0.05, report=False)
        _stream(t, lanes[60_000:], lp, 2.0)
        spawned = destroyed = 0
        for k in range(fill_steps + 10):
            t.step(0.05, report=name != "mesh")
            if name != "mesh":
                spawned += t.last_report["n_spawned"]
                destroyed += t.last_report["n_destroyed"]
        runs[name] = (t.read_agents(), spawned, destroyed)
    a, spawned, destroyed = runs["tiled"]
    print(f"200,000 sinks: {len(a)} agents alive, {spawned} spawned and {destroyed} destroyed after the second batch")
    assert 500_000 < len(a) <= 1_050_000 and destroyed > 100_000
    assert len(np.unique(a["id"])) == len(a)
    assert (spawned, destroyed) == runs["gather"][1:]
    assert a.tobytes() == runs["gather"][0].tobytes() == runs["mesh"][0].tobytes()
    assert np.isfinite(a["x"]).all() and (a["next_waypoint"] == 0).all()
    # a 1,048,576th group, or a source-sink with more waypoints than the narrower counter holds, is refused
    small = Simulation(LocationHash2D(**grid))
    _stream(small, lanes[:66_000], lp, 2.0)
    with pytest.raises(Exception, match="4,095 waypoints"):
        small.add_source_sink(SourceSink((10.0, 10.0), 0.5, MonotonicCrowd(1000.0), StubHighLevelPlan((0.0, 1.0)), lp,
                                         [(10.0, 10.0 + 0.001 * k) for k in range(5000)], False, 2.0))


def test_config4_four_million_agents_with_hotspots():
    """configs[4] at full size: 4M agents, half a uniform background, half in Gaussian hotspots
    (up to 4.9 agents/m^2: neighbour lists beyond 64 entries, windows walked in chunks).  Tiled
    kernel = gather kernel = 4 x 2 tile mesh with weighted cuts, bit for bit; nobody lost,
    duplicated or non-finite; nearly everybody feels a force."""
    n = 4_000_000
    pts, grid, extent, group = scenes.hotspot_crowd(n, seed=7, cell_size=2.0)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)

    def run(target, steps=2):
        scenes.add_counterflow(target, pts, group, scenes.CREEP_SPEED, lp, 2.0)
        for _ in range(steps):
            target.step(0.05, report=False)
        out = target.read_agents()
        rep = None
        if isinstance(target, Simulation):
            target.step(0.05)
            rep = dict(target.last_report)
        return out, rep

    tiled, rep = run(Simulation(LocationHash2D(**grid), flags=2 | _abi.CS_CFG_DENSE, capacity_hint=n + 1024))
    gather, rep_g = run(Simulation(LocationHash2D(**grid), flags=1 | _abi.CS_CFG_DENSE, capacity_hint=n + 1024))
    mesh_t = LocalTileMesh(LocationHash2D(**grid), (4, 2), halo_cells=1, density_per_cell=30.0,
                           flags=_abi.CS_CFG_DENSE, weights=pts)
    counts = mesh_t.layout.tile_counts(pts, LocationHash2D(**grid))
    mesh, _ = run(mesh_t)
    print(f"configs[4]: 4M agents, tiles {counts.reshape(-1).tolist()} (max/mean {counts.max() / counts.mean():.3f}), "
          f"report {rep}")
    assert len(tiled) == n and (tiled["id"] == np.arange(n)).all()
    assert tiled.tobytes() == gather.tobytes()
    assert tiled.tobytes() == mesh.tobytes()
    assert np.isfinite(tiled["x"]).all() and np.isfinite(tiled["vx"]).all()
    assert rep["n_agents"] == n and rep["n_tti_zero"] == rep_g["n_tti_zero"] == 0 and rep["n_nonfinite"] == 0
    force = np.hypot(tiled["vx"], np.abs(tiled["vy"]) - scenes.CREEP_SPEED)
    assert np.mean(force > 0) > 0.9
    assert counts.max() / counts.mean() < 1.2


# ---- long enough to cross the cuts (review of round 3: the full-size runs above make 2-3 steps) ----------------
def test_config2_one_million_walkers_cross_every_cut_for_200_steps():
    """configs[2] as it is benchmarked: the 1M-agent crowd WALKING at 1.3 m/s (6.5 cm per step) on a 4 x 2 mesh, 200
    steps = 13 m: every agent changes cell about six times, tens of thousands migrate over a cut, the ghost rings are
    refilled 200 times.  The mesh behind the C ABI (cs_mesh_*), the Python mesh, the tiled and the gather kernel of one
    engine: the same bits; nobody lost or duplicated; the walkers walked."""
    import bench
    from rmf_crowdsim_amd.tiles import NativeTileMesh
    n, steps = 1_000_000, 200
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, room=bench.walk_room(steps))
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    index = LocationHash2D(**grid)

    def run(target):
        ids = scenes.add_walking_crowd(target, pts, group, lp, 2.0)
        for k in range(steps):
            target.step(0.05, report=False)
        return target.read_agents(), ids

    tiled, ids = run(Simulation(index, flags=2, capacity_hint=n + 1024))
    gather, _ = run(Simulation(index, flags=1, capacity_hint=n + 1024))
    assert len(tiled) == n and (tiled["id"] == np.arange(n)).all()
    assert tiled.tobytes() == gather.tobytes()
    del gather
    native_mesh = NativeTileMesh(index, (4, 2), 1, density_per_cell=15.0, weights=pts, capacity_hint=160_000)
    rects = native_mesh.tile_rects()
    native, _ = run(native_mesh)
    assert tiled.tobytes() == native.tobytes()
    del native, native_mesh
    local, _ = run(LocalTileMesh(index, (4, 2), halo_cells=1, density_per_cell=15.0, weights=pts, capacity_hint=160_000))
    assert tiled.tobytes() == local.tobytes()
    assert np.isfinite(tiled["x"]).all() and np.isfinite(tiled["vx"]).all()
    start = np.empty_like(pts)
    start[np.asarray(ids)] = pts
    walked = tiled["x"] - start[:, 0]
    assert abs(float(walked.mean()) - steps * 0.05 * scenes.WALK_SPEED) < 1e-3 and float(walked.min()) > 12.9
    # agents that crossed an x cut of the mesh (cell rows cx0 of tiles 1..3): thousands per cut
    cuts = sorted(set(int(r[0]) for r in rects if r[0] > 0))
    cut_x = [grid["offset"][0] + c * 2.0 for c in cuts]
    crossed = [int(((start[:, 0] < x) & (tiled["x"] >= x)).sum()) for x in cut_x]
    print(f"configs[2], 200 steps walking: crossed the x cuts at {cut_x}: {crossed}")
    assert len(cuts) == 3 and min(crossed) > 10_000


def test_config2_split_launches_of_the_overlapped_schedule_at_full_size(monkeypatch):
    """configs[2] under CS_CFG_TILE_OVERLAP as far as one GPU can take it through cs_mesh_*: every tile of the 4 x 2 mesh
    steps its BORDER windows as a launch of their own, ahead of the interior ones (CS_TILE_SPLIT=1: the decomposition the
    overlapped schedule runs on; the border launch packs the next step's halo records, the interior launch checks that
    nobody it moved can reach the band), 1M walkers, 200 steps, 21,000 agents over each cut: the single engine's bits.
    (The two streams and the exchange made ahead need a communicator: tests/test_gpu_tiles.py runs them on a tile of
    configs[2]'s 125,000 agents whose peers are the rank itself; real peers need more than one GPU.)"""
    import bench
    from rmf_crowdsim_amd.tiles import NativeTileMesh
    monkeypatch.setenv("CS_TILE_SPLIT", "1")
    n, steps = 1_000_000, 200
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, room=bench.walk_room(steps))
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    index = LocationHash2D(**grid)
    out = []
    for target in (Simulation(index, flags=2, capacity_hint=n + 1024),
                   NativeTileMesh(index, (4, 2), 1, density_per_cell=15.0, weights=pts, capacity_hint=160_000,
                                  flags=_abi.CS_CFG_TILE_OVERLAP)):
        scenes.add_walking_crowd(target, pts, group, lp, 2.0)
        tile0 = target.tile(0) if isinstance(target, NativeTileMesh) else None
        if tile0 is not None:
            tile0.profile_stride(8)
            tile0.profile_enable((1 << _abi.CS_K_STEP_BORDER) | (1 << _abi.CS_K_STEP_INTERIOR))
        for k in range(steps):
            target.step(0.05, report=False)
        out.append(target.read_agents())
        if tile0 is not None:  # the launches really were split
            prof = tile0.profile_read()
            assert prof["step_border"]["launches"] >= 20 and prof["step_interior"]["launches"] >= 20
    assert len(out[0]) == n and out[0].tobytes() == out[1].tobytes()


def test_config4_four_million_agents_50_steps_with_a_recut():
    """configs[4] at full size for 52 steps, the crowd walking (3.4 m: agents leave and enter the hotspots' cells and
    cross the weighted cuts), one re-cut of the running mesh at step 25.  The mesh behind the C ABI == one engine, bit
    for bit, at the re-cut and at the end."""
    from rmf_crowdsim_amd.tiles import NativeTileMesh
    n, steps = 4_000_000, 52
    pts, grid, extent, group = scenes.hotspot_crowd(n, seed=7, cell_size=2.0)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    index = LocationHash2D(**grid)
    single = Simulation(index, flags=2 | _abi.CS_CFG_DENSE, capacity_hint=n + 1024)
    mesh = NativeTileMesh(index, (4, 2), 1, density_per_cell=30.0, flags=_abi.CS_CFG_DENSE, capacity_hint=700_000)  # even cuts first
    for t in (single, mesh):
        scenes.add_walking_crowd(t, pts, group, lp, 2.0, creep=scenes.CREEP_SPEED * 0.1)
    before = mesh.tile_counts()
    for k in range(steps):
        if k == 25:
            assert single.read_agents().tobytes() == mesh.read_agents().tobytes()
            after = mesh.recut()
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    a, b = single.read_agents(), mesh.read_agents()
    print(f"configs[4], {steps} steps walking, re-cut at 25: max/mean {before.max() / before.mean():.3f} -> "
          f"{after.max() / after.mean():.3f}")
    assert len(a) == n and (a["id"] == np.arange(n)).all() and a.tobytes() == b.tobytes()
    assert np.isfinite(a["x"]).all() and np.isfinite(a["vx"]).all()
    assert after.sum() == n and after.max() / after.mean() < 1.05 < 1.12 < before.max() / before.mean()
    single.step(0.05)
    assert single.last_report["n_agents"] == n and single.last_report["n_nonfinite"] == 0


def test_config1_walking_crowd_1000_steps_against_the_f64_path():
    """The default bench workload (the crowd walking at 1.3 m/s, scenes.add_walking_crowd) at configs[1]'s 100,000
    agents for the north star's 1000 steps (65 m: every agent changes cell ~32 times) against the f64 CPU path
    (oracle_fast_steps, bit-identical to the reference-shaped oracle): positions within 1e-4 of the extent."""
    import os
    import bench
    from oracle_sim import fast_steps
    n, steps = 100_000, 1000
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, room=bench.walk_room(steps))
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    speed = min(scenes.CREEP_SPEED, 0.25 / (steps + 2))   # bench.py's rule: at most 2.5 cm of closing over the run
    sim = Simulation(LocationHash2D(**grid))
    ids = np.asarray(scenes.add_walking_crowd(sim, pts, group, lp, 2.0, creep=speed))
    for k in range(steps - 1):
        sim.step(0.05, report=False)
    sim.step(0.05)
    assert sim.last_report["n_tti_zero"] == 0 and sim.last_report["n_nonfinite"] == 0 and sim.last_report["n_agents"] == n
    a = sim.read_agents()
    by_id = np.empty_like(pts)
    by_id[ids] = pts
    pref = np.zeros_like(pts)
    pref[:, 0] = scenes.WALK_SPEED
    pref[ids, 1] = np.where(group == 0, speed, -speed)
    struck = np.zeros(n, dtype=np.uint8)
    xy, vel, sec = fast_steps(by_id, pref, scenes.METRIC_ZANLUNGO, 2.0, grid, 0.05, steps,
                              threads=min(16, os.cpu_count() or 1), spurious=struck)
    ok = np.isfinite(xy).all(axis=1)
    assert np.isfinite(a["x"]).all() and ((~ok) == (struck != 0)).all() and (~ok).sum() <= 60
    dp = np.hypot(a["x"] - xy[:, 0], a["y"] - xy[:, 1])[ok]
    print(f"configs[1] walking, 1000 steps: |dp|/L = {dp.max() / extent:.2e}; {int((~ok).sum())} agents NaN on the "
          f"reference's f64 path; walked {float((xy[ok, 0] - by_id[ok, 0]).mean()):.2f} m; CPU side {sec:.1f} s")
    assert dp.max() / extent <= 1e-4
    assert abs(float((a["x"] - by_id[:, 0]).mean()) - steps * 0.05 * scenes.WALK_SPEED) < 1e-2
