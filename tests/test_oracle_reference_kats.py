"""Pins the CPU oracle against every known answer the reference's own tests hold for the
`Simulation::step` path, plus the hand-derived Zanlungo KATs.

Each test names the reference test it restates.  CPU only.
"""
import math

import numpy as np
import pytest

from oracle_sim import OracleSimulation, load_oracle
from rmf_crowdsim_amd import (EventListener, IdParityHighLevelPlan, LocationHash2D,
                              MonotonicCrowd, NoLocalPlan, SourceSink, StubHighLevelPlan,
                              Zanlungo, CrowdSimError, SeededPoissonCrowd)


# ---- zanlungo.rs:225-236 ----------------------------------------------------
def test_time_to_collision_head_on():
    lib = load_oracle()
    # Zanlungo::new(1, 10, 0, 5, 0.1, 4): agent_radius 4
    assert lib.oracle_time_to_collision(4.0, 1.0, 0.0, -10.0, 0.0) == 6.0


def test_time_to_collision_never_collide():
    lib = load_oracle()
    assert lib.oracle_time_to_collision(4.0, 1.0, 0.0, 10.0, 0.0) == math.inf


def test_time_to_collision_equal_velocities_is_inf():
    # SURVEY §8a a6 [derived]: a = 0 gives 0/0 = NaN, every comparison false -> +inf
    lib = load_oracle()
    assert lib.oracle_time_to_collision(0.5, 0.0, 0.0, 3.0, 0.0) == math.inf


# ---- location_hash_2d.rs:311-397 ---------------------------------------------
def _populated_index():
    sim = OracleSimulation(LocationHash2D(10.0, 10.0, 0.5, (0.0, 0.0)))
    pts = [(x + 0.5, y + 0.5) for x in range(10) for y in range(10)]
    ids = sim.add_agents(pts, StubHighLevelPlan((0, 0)), NoLocalPlan(), 1.0)
    assert ids == list(range(100))
    return sim, np.array(pts)


def test_nearest_neighbours():
    sim, pts = _populated_index()
    assert sim.get_nearest_neighbours(1, (0.6, 0.6)) == [0]
    q = np.array([1.7, 1.6])
    dist = np.sqrt(((pts - q) ** 2).sum(axis=1))
    naive = [int(i) for i in np.argsort(dist, kind="stable")[:4]]
    assert sim.get_nearest_neighbours(4, q) == naive


def test_radius_search():
    sim, pts = _populated_index()
    q = np.array([4.0, 4.0])
    dist = np.sqrt(((pts - q) ** 2).sum(axis=1))
    naive = {int(i) for i in np.nonzero(dist < 1.1)[0]}
    assert set(sim.get_neighbours_in_radius(1.1, q)) == naive


def test_update():
    lib = load_oracle()
    sim = OracleSimulation(LocationHash2D(2.0, 2.0, 1.0, (0.0, 0.0)))
    assert lib.oracle_index_add_or_update(sim._engine, 1, 0.0, 0.0) == 0
    assert sim.get_neighbours_in_radius(1.0, (0.0, 0.0)) == [1]
    assert lib.oracle_index_add_or_update(sim._engine, 1, 1.0, 0.0) == 0
    assert sim.get_neighbours_in_radius(1.0, (0.0, 0.0)) == []  # strict `<`


def test_remove():
    lib = load_oracle()
    sim = OracleSimulation(LocationHash2D(1.0, 1.0, 1.0, (0.0, 0.0)))
    assert lib.oracle_index_add_or_update(sim._engine, 1, 0.0, 0.0) == 0
    assert len(sim.get_neighbours_in_radius(1.1, (0.0, 0.0))) == 1
    lib.oracle_index_remove(sim._engine, 1)
    assert len(sim.get_neighbours_in_radius(1.1, (0.0, 0.0))) == 0


def test_index_out_of_bounds_is_an_error():
    # location_hash_2d.rs:61-63 through add_agents (lib.rs:146-149)
    sim = OracleSimulation(LocationHash2D(2.0, 2.0, 1.0, (0.0, 0.0)))
    with pytest.raises(CrowdSimError, match="Index out of bounds"):
        sim.add_agents([(5.0, 0.5)], StubHighLevelPlan((0, 0)), NoLocalPlan(), 1.0)


def test_negative_coordinates_are_clamped_into_row_zero():
    # `as usize` saturates, so (-3, 0.5) is binned into cell (0, 0)
    sim = OracleSimulation(LocationHash2D(2.0, 2.0, 1.0, (0.0, 0.0)))
    sim.add_agents([(-3.0, 0.5)], StubHighLevelPlan((0, 0)), NoLocalPlan(), 1.0)
    assert sim.get_neighbours_in_radius(3.2, (0.1, 0.5)) == [0]


# ---- lib.rs:423-453 ----------------------------------------------------------
def test_step_integration():
    sim = OracleSimulation(LocationHash2D(1000.0, 1000.0, 20.0, (-500.0, -500.0)))
    assert len(sim.agents) == 0
    ids = sim.add_agents([(0.0, 0.0)], StubHighLevelPlan((1.0, 0.0)), NoLocalPlan(), 100.0)
    assert len(ids) == 1 and len(sim.agents) == 1
    sim.step(1.0)
    assert len(sim.agents) == 1
    assert np.linalg.norm(sim.agents[0].position - np.array([1.0, 0.0])) < 1e-5


# ---- tests/event_listeners_test.rs:65-111 --------------------------------------
class MockEventListener(EventListener):
    def __init__(self):
        self.added, self.removed = [], []

    def agent_spawned(self, position, agent):
        self.added.append(agent)

    def agent_destroyed(self, agent):
        self.removed.append(agent)


def run_event_listener_source_sink_api(sim_cls):
    sim = sim_cls(LocationHash2D(1000.0, 1000.0, 20.0, (-500.0, -500.0)))
    source_sink = SourceSink(source=(0.0, 0.0), waypoints=[(20.0, 0.0)], radius_sink=1.0,
                             crowd_generator=MonotonicCrowd(1.0),
                             high_level_planner=StubHighLevelPlan((1.0, 0.0)),
                             local_planner=NoLocalPlan(), agent_eyesight_range=5.0,
                             loop_forever=False)
    listener = MockEventListener()
    sim.add_event_listener(listener)
    sim.add_source_sink(source_sink)
    for steps in range(20):
        assert len(sim.agents) == steps
        assert len(listener.added) == steps
        sim.step(1.0)
    for steps in range(20, 40):
        assert len(sim.agents) == 20
        assert len(listener.added) == steps
        assert len(listener.removed) == steps - 20
        sim.step(1.0)
    assert listener.removed == list(range(20))
    return sim


def test_event_listener_source_sink_api():
    run_event_listener_source_sink_api(OracleSimulation)


# ---- hand-derived KATs, SURVEY.md §8c -----------------------------------------
def _z1(pj):
    sim = OracleSimulation(LocationHash2D(1000.0, 1000.0, 20.0, (-500.0, -500.0)))
    lp = Zanlungo(1.0, 1.0, 0.0, 1.0, 1.0, 0.5)
    # one warm-up step gives agent 0 velocity (1,0) and agent 1 velocity (0,0) with all
    # TTCs infinite (KAT-Z2: every velocity is 0 on the first step), so start dt = 0.
    sim.add_agents([(0.0, 0.0)], StubHighLevelPlan((1.0, 0.0)), lp, 100.0)
    sim.add_agents([pj], StubHighLevelPlan((0.0, 0.0)), lp, 100.0)
    sim.step(0.0)
    a = sim.agents
    assert tuple(a[0].velocity) == (1.0, 0.0) and tuple(a[1].velocity) == (0.0, 0.0)  # KAT-Z2
    assert tuple(a[0].position) == (0.0, 0.0)
    return sim


def test_kat_z1_perpendicular_avoidance_force():
    sim = _z1((3.0, 0.0))
    sim.step(0.05)
    a = sim.agents
    f = -0.8 * math.exp(0.5)
    assert f == pytest.approx(-1.3189770165601027, rel=1e-15)
    assert a[0].velocity[0] == 1.0
    assert a[0].velocity[1] == pytest.approx(-1.3189770165601027, rel=1e-14)
    assert a[0].position[0] == pytest.approx(0.05, rel=1e-15)
    assert a[0].position[1] == pytest.approx(-0.06594885082800514, rel=1e-14)
    # larger id: weight 0 -> zero force, stays put
    assert tuple(a[1].velocity) == (0.0, 0.0) and tuple(a[1].position) == (3.0, 0.0)
    assert sim.last_report["n_tti_zero"] == 0


def test_kat_z3_overlap_gives_clamped_force_and_nan():
    sim = _z1((0.3, 0.0))
    sim.step(0.05)
    a = sim.agents
    assert sim.last_report["n_tti_zero"] == 2
    # smaller id: magnitude 2*1*1/0 = +inf -> clamped to 1e15, direction perpendicular
    assert abs(a[0].velocity[1]) > 1e14 and math.isfinite(a[0].velocity[1])
    # larger id: 0 * ... / 0 = NaN
    assert math.isnan(a[1].velocity[0]) and math.isnan(a[1].position[0])
    assert sim.last_report["n_nonfinite"] == 1


def test_kat_s1_monotonic_crowd_rounding():
    import datetime
    dt = datetime.timedelta(seconds=0.05)
    assert MonotonicCrowd(9.9).get_number_to_spawn(dt) == 0
    assert MonotonicCrowd(10.0).get_number_to_spawn(dt) == 1   # round(0.5) away from zero
    assert MonotonicCrowd(29.0).get_number_to_spawn(dt) == 1
    assert MonotonicCrowd(30.0).get_number_to_spawn(dt) == 2
    # and the oracle's built-in generator agrees: rate 9.9 never spawns, rate 10 does
    for rate, expect in ((9.9, 0), (10.0, 1)):
        sim = OracleSimulation(LocationHash2D(100.0, 100.0, 5.0, (-50.0, -50.0)))
        sim.add_source_sink(SourceSink((0.0, 0.0), 1.0, MonotonicCrowd(rate),
                                       StubHighLevelPlan((1.0, 0.0)), NoLocalPlan(),
                                       [(20.0, 0.0)], False, 5.0))
        sim.step(0.05)
        assert len(sim.agents) == expect


def test_spawn_blocked_while_source_is_occupied():
    # lib.rs:212-217: hard-coded 0.4 occupancy radius, at most ONE agent per sink per step
    sim = OracleSimulation(LocationHash2D(100.0, 100.0, 5.0, (-50.0, -50.0)))
    sim.add_source_sink(SourceSink((0.0, 0.0), 1.0, MonotonicCrowd(100.0),
                                   StubHighLevelPlan((1.0, 0.0)), NoLocalPlan(),
                                   [(20.0, 0.0)], False, 5.0))
    counts = []
    for _ in range(12):
        sim.step(0.1)   # 0.1 m per step: the source frees up once the last agent is >= 0.4 away
        counts.append(len(sim.agents))
    assert counts == [1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3]


def test_seeded_poisson_is_reproducible():
    def run():
        sim = OracleSimulation(LocationHash2D(100.0, 100.0, 5.0, (-50.0, -50.0)))
        sim.add_source_sink(SourceSink((0.0, 0.0), 1.0, SeededPoissonCrowd(4.0, 1234),
                                       StubHighLevelPlan((2.0, 0.0)), NoLocalPlan(),
                                       [(20.0, 0.0)], False, 5.0))
        out = []
        for _ in range(40):
            sim.step(0.25)
            out.append(len(sim.agents))
        return out
    a, b = run(), run()
    assert a == b and 3 < a[-1] < 40


def test_viz_scene_three_agents_runs_and_stays_finite():
    # rmf_crowdsim_viz/src/main.rs:64-94: the reference's only Zanlungo scene
    sim = OracleSimulation(LocationHash2D(1000.0, 1000.0, 20.0, (-500.0, -500.0)))
    sim.add_agents([(100.0, 100.0), (100.0, -100.0), (60.0, 100.0)],
                   IdParityHighLevelPlan((0.0, 10.0)), Zanlungo(1.0, 1.0, 0.0, 40.0, 2.0, 20.0),
                   100.0)
    for _ in range(200):
        sim.step(0.05)
    arr = sim.read_agents()
    assert np.isfinite(arr["x"]).all() and np.isfinite(arr["vy"]).all()
    # ids 0 and 2 (even) head -y, id 1 heads +y
    assert arr["y"][0] < 100.0 and arr["y"][1] > -100.0


# ---- route follower (the per-step half of RMFPlanner, rmf/mod.rs:195-242) --------------------
class DoglegRoutes:
    """A deterministic stand-in for RMFPlanner::plan_route: start, a point 2 to the left of the
    midpoint, goal.  Counts how often it is asked."""

    def __init__(self):
        self.calls = []

    def __call__(self, start, goal):
        self.calls.append((start, goal))
        if goal[0] > 900.0:
            return None  # "Failed to find contiguous path"
        mx, my = 0.5 * (start[0] + goal[0]), 0.5 * (start[1] + goal[1])
        dx, dy = goal[0] - start[0], goal[1] - start[1]
        n = math.hypot(dx, dy) or 1.0
        return [start, (mx - 2.0 * dy / n, my + 2.0 * dx / n), goal]


def run_route_follower_kat(sim_cls):
    from rmf_crowdsim_amd import RouteFollower
    routes = DoglegRoutes()
    sim = sim_cls(LocationHash2D(100.0, 100.0, 2.0, (-50.0, -50.0)))
    sim.add_source_sink(SourceSink((0.0, 0.0), 1.0, MonotonicCrowd(10.0), RouteFollower(routes, scale=1.0),
                                   NoLocalPlan(), [(10.0, 0.0)], False, 2.0))
    # an agent that never got a target: get_desired_velocity is None, it stays (rmf/mod.rs:211-214)
    idle = sim.add_agents([(-20.0, -20.0)], RouteFollower(routes, scale=1.0), NoLocalPlan(), 2.0)[0]
    sim.step(0.1)
    a = sim.read_agents()
    first = a[a["id"] != idle][0]
    # spawned at the source = route[0]: inside 0.1, so it heads for route[1] = (5, 2) at speed 1
    u = np.array([5.0, 2.0]) / math.hypot(5.0, 2.0)
    assert np.allclose([first["x"], first["y"]], 0.1 * u, atol=1e-6)
    assert np.allclose([first["vx"], first["vy"]], u, atol=1e-6)
    trace = []
    for _ in range(150):
        sim.step(0.1)
        a = sim.read_agents()
        me = a[a["id"] == first["id"]]
        trace.append((float(me["x"][0]), float(me["y"][0])) if len(me) else None)
    alive = [t for t in trace if t is not None]
    # it passes within 0.1 + one stride of the dogleg point, then reaches the sink and is removed
    assert min(math.hypot(x - 5.0, y - 2.0) for x, y in alive) < 0.2
    assert trace[-1] is None and max(x for x, _ in alive) > 8.9
    stay = sim.read_agents()
    stay = stay[stay["id"] == idle][0]
    assert (stay["x"], stay["y"], stay["vx"], stay["vy"]) == (-20.0, -20.0, 0.0, 0.0)
    # one plan for all the agents of this sink: same (start, goal) hash pair (rmf/mod.rs:220-222)
    assert len(routes.calls) == 1 and routes.calls[0] == ((0.0, 0.0), (10.0, 0.0))
    return sim


def test_route_follower_kat():
    run_route_follower_kat(OracleSimulation)


def test_route_follower_keeps_old_route_when_planning_fails():
    from rmf_crowdsim_amd import RouteFollower
    routes = DoglegRoutes()
    sim = OracleSimulation(LocationHash2D(2000.0, 100.0, 2.0, (-50.0, -50.0)))
    sim.add_source_sink(SourceSink((0.0, 0.0), 1.0, MonotonicCrowd(10.0), RouteFollower(routes, speed=2.0),
                                   NoLocalPlan(), [(4.0, 0.0), (1000.0, 0.0)], False, 2.0))
    for _ in range(60):
        sim.step(0.1)
    a = sim.read_agents()
    # the second target cannot be planned: agents keep following the first route to its end (4, 0)
    lead = a[a["id"] == 0][0]
    assert lead["next_waypoint"] == 1 and abs(lead["x"] - 4.0) < 0.3 and abs(lead["y"]) < 0.3


# ---- certifying a parity scene --------------------------------------
def _run_counterflow(gauss_seidel, monkeypatch, speed, n=3000, steps=30):
    from rmf_crowdsim_amd import scenes
    monkeypatch.setenv("CS_ORACLE_GAUSS_SEIDEL", "1" if gauss_seidel else "0")
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=11, cell_size=2.0)
    sim = OracleSimulation(LocationHash2D(**grid))
    scenes.add_counterflow(sim, pts, group, speed, Zanlungo(*scenes.METRIC_ZANLUNGO), 2.0)
    sim.count_shell_crossings(not gauss_seidel)
    crossings = 0
    for _ in range(steps):
        sim.step(0.05)
        crossings += sim.shell_crossings
    return sim.read_agents(), crossings


def test_reference_visiting_order_does_not_matter_in_the_bench_scene(monkeypatch):
    """The reference updates its index inside the agent loop (lib.rs:299), so who counts as a
    neighbour can depend on the visiting order for pairs that sit on the eyesight shell.  The
    oracle counts those pairs; in the creeping counter-flow (bench.py's scene) the few there are
    involve only neighbours without right of way and infinite time to collision, and the
    reference's in-loop update (ascending ids) gives the same bits as the Jacobi form that the
    engine implements.  At ten times the speed the two forms differ, by less than the 1e-4
    parity tolerance."""
    from rmf_crowdsim_amd import scenes
    jac, crossings = _run_counterflow(False, monkeypatch, scenes.CREEP_SPEED)
    gs, _ = _run_counterflow(True, monkeypatch, scenes.CREEP_SPEED)
    assert 0 < crossings < 0.01 * 3000 * 30
    assert jac.tobytes() == gs.tobytes()
    jac, crossings_fast = _run_counterflow(False, monkeypatch, 0.01)
    gs, _ = _run_counterflow(True, monkeypatch, 0.01)
    assert crossings_fast > crossings
    ok = np.isfinite(jac["x"]) & np.isfinite(gs["x"])  # the model's own NaN agents aside
    err = max(np.abs(jac["x"] - gs["x"])[ok].max(), np.abs(jac["y"] - gs["y"])[ok].max()) / 60.0
    assert ok.mean() > 0.99 and 0.0 < err < 1e-4


def test_openmp_baseline_reproduces_the_oracle_bit_for_bit():
    """bench.py's second CPU column (cell-sorted arrays, OpenMP) runs the oracle's own Zanlungo
    arithmetic in the oracle's order: identical results, on one thread and on several."""
    from oracle_sim import fast_steps
    from rmf_crowdsim_amd import scenes
    n = 3000
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=11, cell_size=2.0)
    sim = OracleSimulation(LocationHash2D(**grid))
    ids = scenes.add_counterflow(sim, pts, group, scenes.CREEP_SPEED, Zanlungo(*scenes.METRIC_ZANLUNGO), 2.0)
    for _ in range(12):
        sim.step(0.05)
    a = sim.read_agents()
    # the baseline numbers agents in array order: feed it the points in id order
    order = np.argsort(ids)
    pref = np.zeros((n, 2))
    pref[:, 1] = np.where(group[order] == 0, scenes.CREEP_SPEED, -scenes.CREEP_SPEED)
    for threads in (1, 4):
        xy, vel, sec = fast_steps(pts[order], pref, scenes.METRIC_ZANLUNGO, 2.0, grid, 0.05, 12, threads=threads)
        assert sec > 0
        assert np.array_equal(xy[:, 0], a["x"]) and np.array_equal(xy[:, 1], a["y"])
        assert np.array_equal(vel[:, 0], a["vx"]) and np.array_equal(vel[:, 1], a["vy"])


def test_walking_scene_is_certified_on_the_oracle():
    """bench.py's default workload (scenes.add_walking_crowd) on the f64 oracle for the full 1000
    steps of the north star: nobody's time to collision reaches 0, nobody goes non-finite, and the
    velocities stay the preferred ones exactly (the force term underflows, see the scene's
    docstring), so the crowd walks 65 m."""
    import numpy as np
    from rmf_crowdsim_amd import LocationHash2D, Zanlungo, scenes
    n = 1500
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=7, cell_size=2.0, room=70.0)
    sim = OracleSimulation(LocationHash2D(**grid))
    ids = scenes.add_walking_crowd(sim, pts, group, Zanlungo(*scenes.METRIC_ZANLUNGO), 2.0)
    tz = 0
    for _ in range(1000):
        sim.step(0.05)
        tz += sim.last_report["n_tti_zero"] + sim.last_report["n_nonfinite"]
    a = sim.read_agents()
    assert tz == 0 and len(a) == n
    assert (a["vx"] == scenes.WALK_SPEED).all() and (np.abs(a["vy"]) == scenes.CREEP_SPEED).all()
    start = np.empty_like(pts)
    start[ids] = pts
    assert np.allclose(a["x"] - start[:, 0], 65.0, atol=1e-9)


def test_a_foreign_spatial_index_is_refused_with_a_reason():
    """Simulation<T: SpatialIndex> (lib.rs:69; spatial_index.rs:4-14): the mirror has the trait; an index that wraps a
    LocationHash2D is accepted through device_form(), one that does not describe itself as a grid is refused."""
    from rmf_crowdsim_amd import SpatialIndex

    class Wrapped(SpatialIndex):
        def __init__(self):
            self.grid = LocationHash2D(20.0, 20.0, 2.0, (0.0, 0.0))

        def device_form(self):
            return self.grid

    class Foreign(SpatialIndex):
        pass

    sim = OracleSimulation(Wrapped())
    sim.add_agents([(1.0, 1.0)], StubHighLevelPlan((1.0, 0.0)), NoLocalPlan(), 1.0)
    sim.step(1.0)
    assert sim.agents[0].position[0] == pytest.approx(2.0)
    with pytest.raises(Exception, match="does not describe itself as a uniform grid"):
        OracleSimulation(Foreign())
