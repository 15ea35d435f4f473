"""A LocalPlanner that is host code (trait LocalPlanner, local_planners/local_planner.rs:7-18; called at
lib.rs:276-291): the slow path of SURVEY.md section 8b, cs_register_lp_callback.

CPU: through the ORACLE's implementation of the same C-ABI entry point, the reference's Zanlungo restated a
second time in numpy (oracle/zanlungo_restatement.py) and plugged in as a user planner gives the trajectory
of the oracle's own Zanlungo: the callback hands a planner exactly what the reference hands it (the agent
with its recommended velocity, its neighbours' old states, itself excluded).
GPU: the engine's slow path against that: the same numpy planner plugged into the ENGINE gives the oracle's
Zanlungo trajectory to the path's tolerance (1e-4 of L); a planner of no physical meaning that looks at every
field it is handed gives the same trajectory on engine and oracle; remove_agent reaches the planner.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zanlungo_restatement as zr  # noqa: E402
from oracle_sim import OracleSimulation  # noqa: E402
from rmf_crowdsim_amd import (IdParityHighLevelPlan, LocalPlanner, LocationHash2D, MonotonicCrowd, NoLocalPlan,  # noqa: E402
                              Simulation, SourceSink, StubHighLevelPlan, Zanlungo, scenes)

PARAMS = (1.0, 1.0, 0.0, 0.4, 2.0, 0.2)


class NumpyZanlungo(LocalPlanner):
    """zanlungo.rs:23-217 on the host, through the second restatement."""

    def __init__(self, *params):
        self.z = zr.Zanlungo(*params)
        self.calls = 0

    def get_desired_velocity(self, agent, nearby_agents, recommended_velocity):
        self.calls += 1
        me = zr.Agent(agent.agent_id, agent.position, agent.velocity, recommended_velocity)
        nearby = [zr.Agent(a.agent_id, a.position, a.velocity, (0.0, 0.0)) for a in nearby_agents]
        return self.z.get_desired_velocity(me, nearby, recommended_velocity)


class Nonsense(LocalPlanner):
    """Looks at everything it is given: ids, positions, velocities, the order of the neighbours, the waypoint
    counter, the eyesight; returns something bounded."""

    def __init__(self):
        self.removed = []

    def get_desired_velocity(self, agent, nearby_agents, recommended_velocity):
        assert agent.agent_id not in [a.agent_id for a in nearby_agents]          # lib.rs:284
        assert np.allclose(agent.preferred_vel, recommended_velocity)
        w = np.array(recommended_velocity, dtype=np.float64)
        for rank, a in enumerate(nearby_agents):
            d = a.position - agent.position
            assert np.hypot(*d) < agent.eyesight_range                             # location_hash_2d.rs:251
            w += 1e-3 * (rank + 1) * np.array([-d[1], d[0]]) + 1e-2 * a.velocity * ((a.agent_id % 3) - 1)
        w[0] += 1e-3 * agent.next_waypoint
        return np.clip(w, -0.5, 0.5)

    def remove_agent(self, agent_id):
        self.removed.append(agent_id)


def _counterflow(cls, lp, n=400, steps=30, **kw):
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=3, cell_size=2.0)
    sim = cls(LocationHash2D(**grid), **kw)
    scenes.add_counterflow(sim, pts, group, scenes.CREEP_SPEED, lp, 2.0)
    for _ in range(steps):
        sim.step(0.05)
    return sim.read_agents(), extent, sim


def _err(a, b, extent):
    """max |dp| / L over the agents that are finite on both sides (the reference's f64 arithmetic loses about
    one agent per 1e6 agent-steps of this scene to 0/0, DESIGN.md section 5: at most a handful here)"""
    assert (a["id"] == b["id"]).all()
    ok = np.isfinite(a["x"]) & np.isfinite(b["x"])
    assert (~ok).sum() <= 3
    return float(np.hypot(a["x"] - b["x"], a["y"] - b["y"])[ok].max() / extent)


def test_numpy_zanlungo_as_a_user_planner_is_the_oracles_zanlungo():
    mine = NumpyZanlungo(*PARAMS)
    a, extent, _ = _counterflow(OracleSimulation, mine, n=150, steps=12)
    b, _, _ = _counterflow(OracleSimulation, Zanlungo(*PARAMS), n=150, steps=12)
    assert mine.calls == 150 * 12
    assert _err(a, b, extent) < 1e-12
    assert np.nanmax(np.abs(a["vx"])) > 1e-5  # forces were at work (the counter-flow is in y)


def test_a_local_planner_must_override_get_desired_velocity():
    sim = OracleSimulation(LocationHash2D(20.0, 20.0, 2.0, (0.0, 0.0)))
    with pytest.raises(Exception, match="override get_desired_velocity"):
        sim.add_agents([(1.0, 1.0)], StubHighLevelPlan((0.0, 0.0)), LocalPlanner(), 1.0)


@pytest.mark.gpu
def test_engine_with_numpy_zanlungo_follows_the_oracles_zanlungo():
    mine = NumpyZanlungo(*PARAMS)
    a, extent, _ = _counterflow(Simulation, mine)
    b, _, _ = _counterflow(OracleSimulation, Zanlungo(*PARAMS))
    c, _, _ = _counterflow(Simulation, Zanlungo(*PARAMS))
    assert mine.calls == 400 * 30
    assert _err(a, b, extent) < 1e-4          # the path's tolerance (BASELINE.json north_star)
    assert _err(a, c, extent) < 1e-4          # and the device planner's trajectory
    assert np.nanmax(np.abs(b["vx"])) > 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 2], ids=["gather", "tiled"])
def test_a_custom_planner_gives_the_same_trajectory_on_engine_and_oracle(kernel):
    out = []
    for cls, kw in ((Simulation, {"flags": kernel}), (OracleSimulation, {})):
        lp = Nonsense()
        pts = scenes.jittered_lattice(3000, 0.63, (10.0, 10.0), 0.2, 5)   # 2.5 agents / m^2 on [10, 45)^2
        extent = 35.0
        sim = cls(LocationHash2D(80.0, 80.0, 2.0, (0.0, 0.0)), **kw)
        ids = sim.add_agents(pts[:2000], IdParityHighLevelPlan((0.0, 0.05)), lp, 1.5)
        sim.add_agents(pts[2000:], StubHighLevelPlan((0.03, 0.0)), Zanlungo(*PARAMS), 2.0)   # device planner beside it
        # a stream of the host planner's agents: one leaves the source whenever the last one is 0.4 m away, and is
        # destroyed two steps later (remove_agent reaches the planner)
        sim.add_source_sink(SourceSink((60.0, 60.0), 0.5, MonotonicCrowd(20.0), StubHighLevelPlan((2.0, 0.0)), lp,
                                       [(60.6, 60.0)], False, 1.5))
        for k in range(25):
            sim.step(0.05)
            if k == 10:
                sim.remove_agents(ids[7])
        out.append((sim.read_agents(), lp.removed, extent))
    (a, removed_a, extent), (b, removed_b, _) = out
    assert len(a) == len(b) >= 3000 and _err(a, b, extent) < 1e-6
    assert float(np.hypot(a["vx"] - b["vx"], a["vy"] - b["vy"]).max()) < 1e-5
    assert removed_a == removed_b and 7 in removed_a and len(removed_a) > 3   # the removal + agents that reached the sink


class Raises(LocalPlanner):
    def get_desired_velocity(self, agent, nearby_agents, recommended_velocity):
        if agent.agent_id == 2:
            raise ValueError("no answer for agent 2")
        return recommended_velocity


def _failing_planner(cls):
    sim = cls(LocationHash2D(20.0, 20.0, 2.0, (0.0, 0.0)))
    sim.add_agents([(1.0, 1.0), (3.0, 1.0), (5.0, 1.0)], StubHighLevelPlan((0.1, 0.0)), Raises(), 1.0)
    with pytest.raises(Exception, match="host LocalPlanner failed") as info:
        sim.step(0.05)
    assert "no answer for agent 2" in str(info.value)   # the planner's own exception is named as the cause
    return sim


def test_a_host_planner_that_raises_fails_the_step_on_the_oracle():
    """An exception inside the ctypes thunk used to be swallowed and the zero-initialised answers applied (the agents
    silently stopped): the callback now reports failure and the step returns Err."""
    _failing_planner(OracleSimulation)


@pytest.mark.gpu
def test_a_host_planner_that_raises_fails_the_step_on_the_engine():
    sim = _failing_planner(Simulation)
    a = sim.read_agents()
    assert (a["x"] == [1.0, 3.0, 5.0]).all() and (a["vx"] == 0.0).all()   # nothing was committed


def _mesh_scene(target, lp):
    pts = scenes.jittered_lattice(3000, 0.63, (10.0, 10.0), 0.2, 5)   # 2.5 agents / m^2 on [10, 45)^2: all four tiles of a 2 x 2 mesh
    ids = target.add_agents(pts[:2000], IdParityHighLevelPlan((0.0, 0.05)), lp, 1.5)
    target.add_agents(pts[2000:], StubHighLevelPlan((0.03, 0.0)), Zanlungo(*PARAMS), 2.0)
    target.add_source_sink(SourceSink((60.0, 60.0), 0.5, MonotonicCrowd(20.0), StubHighLevelPlan((2.0, 0.0)), lp,
                                      [(60.6, 60.0)], False, 1.5))
    return ids


def test_a_host_planner_on_the_oracles_mesh_binding():
    """cs_mesh_register_lp_callback through NativeTileMesh on the oracle's one-process "mesh" (CPU plumbing)."""
    from oracle_sim import load_oracle
    from rmf_crowdsim_amd.tiles import NativeTileMesh
    grid = dict(width=80.0, height=80.0, cell_size=2.0, offset=(0.0, 0.0))
    out = []
    for make in (lambda: OracleSimulation(LocationHash2D(**grid)),
                 lambda: NativeTileMesh(LocationHash2D(**grid), (2, 2), 1, library=load_oracle("f64"))):
        t, lp = make(), Nonsense()
        _mesh_scene(t, lp)
        for k in range(6):
            t.step(0.05)
        out.append((t.read_agents(), lp.removed))
    assert out[0][0].tobytes() == out[1][0].tobytes() and out[0][1] == out[1][1] and len(out[0][1]) > 0


@pytest.mark.gpu
def test_a_host_planner_on_a_tile_mesh_equals_the_single_engine():
    """A `LocalPlanner` in host code on a 2 x 2 mesh (round 4: cs_mesh_register_lp_callback): every tile asks the planner
    for the agents it owns, with ghosts among the neighbours, in canonical order; agents cross the cuts; a source-sink
    of the host planner spawns and destroys (remove_agent reaches the planner).  Same bits as one engine."""
    from rmf_crowdsim_amd.tiles import NativeTileMesh
    grid = dict(width=80.0, height=80.0, cell_size=2.0, offset=(0.0, 0.0))
    out = []
    for make in (lambda: Simulation(LocationHash2D(**grid)), lambda: NativeTileMesh(LocationHash2D(**grid), (2, 2), 1)):
        t, lp = make(), Nonsense()
        ids = _mesh_scene(t, lp)
        for k in range(25):
            t.step(0.05)
            if k == 10:
                t.remove_agents(ids[7])
        out.append((t.read_agents(), sorted(lp.removed)))
    (a, removed_a), (b, removed_b) = out
    assert len(a) >= 3000 and a.tobytes() == b.tobytes()
    assert removed_a == removed_b and 7 in removed_a and len(removed_a) > 3
