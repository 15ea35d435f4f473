"""tests/golden/zanlungo_kats.json: hand-derived known answers for the Zanlungo force (the reference holds no vector
for compute_agent_force / right_of_way_vel / slerp: SURVEY.md section 8c), checked on

  * the numpy restatement of zanlungo.rs (oracle/zanlungo_restatement.py),
  * the C++ oracle: its planner probes, and whole `Simulation::step`s of the scene cases,
  * the device (-m gpu), for the cases Simulation::step can reach (a neighbour's preferred_vel is (0,0) there, so the
    moving-neighbour branch zanlungo.rs:126-139, cases Z5 / Z6, is reachable through the planner alone).

The table is data worked out by hand (tests/golden/make_zanlungo_kats.py holds the arithmetic as closed forms and
regenerates the file; it imports nothing of the oracle or the engine)."""
import ctypes
import json
import math
import os
import sys

import numpy as np
import pytest

from oracle_sim import OracleSimulation, load_oracle
from rmf_crowdsim_amd import HighLevelPlanner, LocationHash2D, Simulation, Zanlungo

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import zanlungo_restatement as zr  # noqa: E402

DP = ctypes.POINTER(ctypes.c_double)


def _table():
    def dec(o):
        if o == "NaN":
            return float("nan")
        if isinstance(o, dict):
            return {k: dec(v) for k, v in o.items()}
        if isinstance(o, list):
            return [dec(v) for v in o]
        return o
    with open(os.path.join(HERE, "golden", "zanlungo_kats.json")) as f:
        return dec(json.load(f))["cases"]


CASES = _table()


def _close(got, want, rel):
    if math.isnan(want):
        return math.isnan(got)
    if want == 0.0:
        return got == 0.0
    return abs(got - want) <= rel * abs(want)


def _rec(a, pref=None):
    p = a["pref"] if pref is None else pref
    return np.array([a["id"], a["p"][0], a["p"][1], a["v"][0], a["v"][1], p[0], p[1]], dtype=np.float64)


def _neighbours(case, me):
    """Everybody else within eyesight (strict <), in the canonical order of the neighbour list: cells x-major /
    y-minor (location_hash_2d.rs:245-246), ascending id inside a cell; preferred_vel (0, 0) as Simulation::step
    hands neighbours over (lib.rs:140,285)."""
    w, h, cs, ox, oy = case["grid"]
    out = []
    for a in case["agents"]:
        if a["id"] == me["id"]:
            continue
        if math.hypot(a["p"][0] - me["p"][0], a["p"][1] - me["p"][1]) < case["eyesight"]:
            cell = (math.floor((a["p"][0] - ox) / cs), math.floor((a["p"][1] - oy) / cs))
            out.append((cell, a["id"], a))
    return [a for _, _, a in sorted(out, key=lambda t: (t[0], t[1]))]


def test_the_table_is_what_its_generator_writes():
    """The committed json is the output of the closed forms in make_zanlungo_kats.py (nobody edited numbers by hand)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_zanlungo_kats", os.path.join(HERE, "golden", "make_zanlungo_kats.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fresh = mod.cases()
    assert [c["name"] for c in fresh] == [c["name"] for c in CASES] == ["Z4", "Z5", "Z6", "Z7", "Z7b", "Z8a", "Z8b", "Z9"]
    for a, b in zip(fresh, CASES):
        for key, want in a["expect"].items():
            got = b["expect"][key]
            flat = lambda v: np.asarray(v["velocity"] if isinstance(v, dict) else v, dtype=float)  # noqa: E731
            assert np.array_equal(flat(want), flat(got), equal_nan=True), a["name"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_restatement_and_oracle_probes(case):
    lib = load_oracle("f64")
    params = np.array(case["planner"], dtype=np.float64)
    z = zr.Zanlungo(*case["planner"])
    rel = case.get("rel", 0.0)
    out = np.zeros(2)
    if case["kind"] == "pair":
        me, other = case["me"], case["other"]
        with np.errstate(all="ignore"):
            f = z.compute_agent_force(zr.Agent(me["id"], me["p"], me["v"], me["pref"]),
                                      zr.Agent(other["id"], other["p"], other["v"], other["pref"]), case["t_i"])
        lib.oracle_zanlungo_pair_force(params.ctypes.data_as(DP), _rec(me).ctypes.data_as(DP), _rec(other).ctypes.data_as(DP),
                                       case["t_i"], out.ctypes.data_as(DP))
        for k in range(2):
            assert _close(float(f[k]), case["expect"]["force"][k], rel), (case["name"], "restatement", f)
            assert _close(float(out[k]), case["expect"]["force"][k], rel), (case["name"], "oracle", out)
        return
    for key, want in case["expect"].items():
        me = next(a for a in case["agents"] if a["id"] == int(key))
        nb = _neighbours(case, me)
        with np.errstate(all="ignore"):
            a = zr.Agent(me["id"], me["p"], me["v"], me["pref"])
            others = [zr.Agent(o["id"], o["p"], o["v"], (0.0, 0.0)) for o in nb]
            t_r = z.compute_tti(a, others)
            v_r = z.get_desired_velocity(a, others, me["pref"])
        buf = np.ascontiguousarray(np.array([_rec(o, (0.0, 0.0)) for o in nb]).reshape(len(nb), 7))
        t_o = lib.oracle_zanlungo_desired_velocity(params.ctypes.data_as(DP), _rec(me).ctypes.data_as(DP), buf.ctypes.data_as(DP),
                                                   len(nb), me["pref"][0], me["pref"][1], out.ctypes.data_as(DP))
        t_rel = case.get("t_rel", 0.0)
        assert _close(float(t_r), want["t_i"], t_rel) and _close(float(t_o), want["t_i"], t_rel), (case["name"], t_r, t_o)
        for k in range(2):
            assert _close(float(v_r[k]), want["velocity"][k], rel), (case["name"], "restatement", key, v_r)
            assert _close(float(out[k]), want["velocity"][k], rel), (case["name"], "oracle", key, out)
        if "terms_in_order" in want:  # the single terms, and that the listed order is the order of the sum
            terms = []
            for o in others:
                with np.errstate(all="ignore"):
                    terms.append(z.compute_agent_force(a, o, t_r))
            for got, exp in zip(terms, want["terms_in_order"]):
                assert _close(float(got[0]), exp[0], 1e-14) and _close(float(got[1]), exp[1], 1e-14)
            assert [o.agent_id for o in others] == [3, 1, 2]


class Scripted(HighLevelPlanner):
    """Gives every agent its velocity of the table in a first step of zero duration (all velocities are 0 then, so every
    time to collision is infinite and the planner returns the recommendation: KAT-Z2), and its preferred velocity from
    then on: the state a scene case describes is reached through Simulation::step alone."""

    def __init__(self, case):
        self.first = {a["id"]: a["v"] for a in case["agents"]}
        self.then = {a["id"]: a["pref"] for a in case["agents"]}
        self.primed = set()

    def get_desired_velocity(self, agent, time):
        if agent.agent_id not in self.primed:
            self.primed.add(agent.agent_id)
            return tuple(self.first[agent.agent_id])
        return tuple(self.then[agent.agent_id])


def _step_scene(cls, case):
    w, h, cs, ox, oy = case["grid"]
    sim = cls(LocationHash2D(w, h, cs, (ox, oy)))
    hlp, lp = Scripted(case), Zanlungo(*case["planner"])
    ids = sim.add_agents([a["p"] for a in case["agents"]], hlp, lp, case["eyesight"])
    assert ids == [a["id"] for a in case["agents"]]
    sim.step(0.0)
    a = sim.read_agents()
    for k, ag in enumerate(case["agents"]):  # the scene as described (the device stores cell-relative f32 offsets)
        assert (a["vx"][k], a["vy"][k]) == tuple(ag["v"])
        assert abs(a["x"][k] - ag["p"][0]) <= 2e-6 and abs(a["y"][k] - ag["p"][1]) <= 2e-6
    sim.step(0.05)
    return sim.read_agents(), sim.last_report


SCENES = [c for c in CASES if c["kind"] == "scene"]


@pytest.mark.parametrize("case", SCENES, ids=[c["name"] for c in SCENES])
def test_scene_cases_through_the_oracles_step(case):
    got, rep = _step_scene(OracleSimulation, case)
    for key, want in case["expect"].items():
        k = int(key)
        for c, name in enumerate(("vx", "vy")):
            assert _close(float(got[name][k]), want["velocity"][c], case.get("rel", 0.0)), (case["name"], key, got[k])
        me = case["agents"][k]
        if not math.isnan(want["velocity"][0]):  # integrated with the new velocity (lib.rs:295-297)
            assert got["x"][k] == pytest.approx(me["p"][0] + 0.05 * want["velocity"][0], abs=1e-15)
            assert got["y"][k] == pytest.approx(me["p"][1] + 0.05 * want["velocity"][1], abs=1e-15)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in SCENES if c["device"]], ids=[c["name"] for c in SCENES if c["device"]])
def test_scene_cases_on_the_device(case):
    """f32 on the device: 2e-6 relative on the forced component; exact zeros, exact recommendations and NaNs as such."""
    got, rep = _step_scene(Simulation, case)
    for key, want in case["expect"].items():
        k = int(key)
        for c, name in enumerate(("vx", "vy")):
            w = want["velocity"][c]
            g = float(got[name][k])
            if math.isnan(w):
                assert math.isnan(g), (case["name"], key, got[k])
            elif w == 0.0 or float(np.float32(w)) == w:
                assert g == w, (case["name"], key, name, g, w)   # untouched components are exact
            else:
                assert abs(g - w) <= 2e-6 * abs(w), (case["name"], key, name, g, w)
