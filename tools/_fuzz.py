import sys, math
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from rmf_crowdsim_amd import *
from rmf_crowdsim_amd.simulation import HighLevelPlanner
from oracle_sim import OracleSimulation
from test_oracle_reference_kats import MockEventListener

class Wander(HighLevelPlanner):
    """host planner: walks towards its target at `speed`; None (stay) for some ids; logs calls"""
    def __init__(self, speed, lazy_mod):
        self.speed, self.lazy_mod = speed, lazy_mod
        self.targets, self.log = {}, []
    def get_desired_velocity(self, agent, time):
        if self.lazy_mod and agent.agent_id % self.lazy_mod == 0:
            return None
        t = self.targets.get(agent.agent_id)
        if t is None:
            return (0.0, 0.0)
        d = np.array(t) - np.array(agent.position)
        n = float(np.linalg.norm(d))
        return tuple(d / n * self.speed) if n > 1e-9 else (0.0, 0.0)
    def set_target(self, agent, point, tolerance):
        self.targets[agent.agent_id] = (float(point[0]), float(point[1])); self.log.append(("set", agent.agent_id, round(float(point[0]), 6), round(float(point[1]), 6), round(float(tolerance[0]), 6)))
    def remove_agent_id(self, agent_id):
        self.targets.pop(agent_id, None); self.log.append(("rm", agent_id))

def scene(t, rng_seed):
    rng = np.random.default_rng(rng_seed)
    hlps = [Wander(float(rng.uniform(0.5, 1.5)), int(rng.choice([0, 3, 5]))) for _ in range(int(rng.integers(1, 3)))]
    stub = StubHighLevelPlan((0.3, -0.2))
    for k in range(int(rng.integers(2, 8))):
        src = rng.uniform(10.0, 70.0, size=2)
        wps = [tuple(rng.uniform(8.0, 72.0, size=2)) for _ in range(int(rng.integers(1, 4)))]
        gen = MonotonicCrowd(float(rng.uniform(0.5, 8.0))) if rng.random() < 0.5 else SeededPoissonCrowd(float(rng.uniform(0.5, 4.0)), int(rng.integers(1, 1 << 30)))
        h = hlps[int(rng.integers(0, len(hlps)))] if rng.random() < 0.8 else stub
        t.add_source_sink(SourceSink(tuple(src), float(rng.uniform(0.4, 2.0)), gen, h, NoLocalPlan(), wps, bool(rng.random() < 0.3), float(rng.uniform(1.0, 3.0))))
    pts = rng.uniform(15.0, 65.0, size=(int(rng.integers(0, 40)), 2))
    if len(pts): t.add_agents(pts, hlps[0], NoLocalPlan(), 2.0)
    return hlps

bad = []
for seed in (range(int(sys.argv[1])) if len(sys.argv) > 1 else range(40)):
    try:
        cell = float([1.0, 2.0, 2.5][seed % 3])
        grid = dict(width=80.0, height=80.0, cell_size=cell, offset=(0.0, 0.0))
        sims = [Simulation(LocationHash2D(**grid)), OracleSimulation(LocationHash2D(**grid))]
        ls, hs = [], []
        for t in sims:
            hs.append(scene(t, 3900 + seed))
            l = MockEventListener(); t.add_event_listener(l); ls.append(l)
        rng = np.random.default_rng(seed)
        for k in range(150):
            for t in sims: t.step(0.1)
            assert len(sims[0]) == len(sims[1]), ("len", k, len(sims[0]), len(sims[1]))
            if rng.random() < 0.1 and len(sims[0]):
                a = sims[0].read_agents(); v = int(a["id"][int(rng.integers(0, len(a)))])
                for t in sims: t.remove_agents(v)
        a, b = sims[0].read_agents(), sims[1].read_agents()
        assert ls[0].added == ls[1].added and ls[0].removed == ls[1].removed, "events"
        assert (a["id"] == b["id"]).all() and (a["next_waypoint"] == b["next_waypoint"]).all()
        err = float(np.hypot(a["x"] - b["x"], a["y"] - b["y"]).max() / 80.0) if len(a) else 0.0
        assert err <= 1e-4, err
        for x, y in zip(hs[0], hs[1]):
            assert x.log == y.log, ("planner log", len(x.log), len(y.log), [p for p in zip(x.log, y.log) if p[0] != p[1]][:3])
        print("ok", seed, "alive", len(a), "spawned", len(ls[0].added), "destroyed", len(ls[0].removed), "log", sum(len(x.log) for x in hs[0]), "err %.1e" % err, flush=True)
    except AssertionError as e:
        bad.append(seed); print("FAIL", seed, str(e)[:400], flush=True)
    except Exception as e:
        import traceback
        bad.append(seed); print("ERROR", seed, type(e).__name__, str(e)[:300], traceback.format_exc()[-500:], flush=True)
print("failed seeds:", bad)
