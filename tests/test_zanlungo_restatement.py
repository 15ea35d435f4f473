"""The C++ oracle's Zanlungo planner against a second, independent restatement of
zanlungo.rs (oracle/zanlungo_restatement.py, written from the Rust alone).  The reference
holds no vector for the force, so this is what stands between a misreading of
zanlungo.rs:93-198 and every parity number: two separate readings must agree, including on
the moving-neighbour branch (:126-139) that Simulation::step never reaches (a neighbour's
preferred_vel is always (0,0) there) and on the NaN / inf / clamp cases.  CPU only."""
import ctypes
import os
import sys

import numpy as np
import pytest

from oracle_sim import load_oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import zanlungo_restatement as zr  # noqa: E402

DP = ctypes.POINTER(ctypes.c_double)


def _ptr(a):
    return a.ctypes.data_as(DP)


def _same(a, b, rtol=1e-12):
    a, b = float(a), float(b)
    if np.isnan(a) or np.isnan(b):
        return np.isnan(a) and np.isnan(b)
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= rtol * max(abs(a), abs(b), 1e-300)


def _random_record(rng, ids, moving, spread=3.0):
    """{id, px, py, vx, vy, pref_x, pref_y}; the kinds of state the branches distinguish."""
    rec = np.zeros(7)
    rec[0] = ids
    rec[1:3] = rng.uniform(-spread, spread, 2)
    kind = rng.integers(0, 8)
    rec[3:5] = (0.0, 0.0) if kind == 0 else rng.normal(0.0, 1.3, 2)
    if moving:
        rec[5:7] = rng.normal(0.0, 1.3, 2) if kind != 1 else rng.normal(0.0, 3e-5, 2)  # around the 1e-4 test
    return rec


def test_reference_fixtures_on_the_restatement():
    """zanlungo.rs:225-236, and SURVEY.md KAT-Z1 worked by hand from the Rust."""
    z = zr.Zanlungo(1, 10, 0, 5, 0.1, 4)
    assert z.time_to_collision((1.0, 0.0), (-10.0, 0.0)) == 6.0
    assert z.time_to_collision((1.0, 0.0), (10.0, 0.0)) == np.inf
    z = zr.Zanlungo(1, 1, 0, 1, 1, 0.5)
    i = zr.Agent(0, (0, 0), (1, 0), (1, 0))
    j = zr.Agent(1, (3, 0), (0, 0), (0, 0))
    assert z.compute_tti(i, [j]) == 2.5
    v = z.get_desired_velocity(i, [j], (1.0, 0.0))
    assert v[0] == 1.0 and v[1] == pytest.approx(-1.3189770165601027, rel=1e-15)
    assert tuple(z.get_desired_velocity(j, [i], (0.0, 0.0))) == (0.0, 0.0)


def test_pair_force_agrees_on_random_pairs():
    """compute_agent_force + right_of_way_vel + slerp on 12,000 random ordered pairs: stationary and
    moving neighbours, either priority order, equal priorities, t_i from 0 through tiny to large."""
    lib = load_oracle("f64")
    rng = np.random.default_rng(20261004)
    n_nan = n_moving = n_forward = n_clamped = 0
    out = np.zeros(2)
    for k in range(12000):
        params = np.array([rng.uniform(0.05, 3.0), 1.0, 0.0, rng.uniform(0.1, 2.0), rng.uniform(0.5, 3.0),
                           rng.uniform(0.05, 0.6)])
        ida, idb = rng.integers(0, 50, 2)
        if k % 97 == 0:
            idb = ida  # equal priorities: right_of_way == 0, weight 1
        moving = k % 2 == 1
        me, other = _random_record(rng, ida, True), _random_record(rng, idb, moving)
        t_i = [0.0, 1e-300, 1e-17, rng.uniform(0.0, 5.0), rng.uniform(0.0, 5.0), rng.uniform(0.0, 50.0)][k % 6]
        lib.oracle_zanlungo_pair_force(_ptr(params), _ptr(me), _ptr(other), t_i, _ptr(out))
        z = zr.Zanlungo(*params)
        with np.errstate(all="ignore"):
            f = z.compute_agent_force(zr.Agent(me[0], me[1:3], me[3:5], me[5:7]),
                                      zr.Agent(other[0], other[1:3], other[3:5], other[5:7]), t_i)
        assert _same(out[0], f[0]) and _same(out[1], f[1]), (k, params, me, other, t_i, out, f)
        n_nan += int(np.isnan(out[0]))
        n_moving += int(ida < idb and np.hypot(other[5], other[6]) >= 1e-4)
        n_forward += int(ida < idb)
        n_clamped += int(np.hypot(out[0], out[1]) > 1e14)
    # every branch was really visited
    assert n_forward > 4000 and n_moving > 1500 and n_nan > 50 and n_clamped > 50


def test_desired_velocity_agrees_on_random_neighbourhoods():
    """get_desired_velocity + compute_tti (min over every neighbour, forces in list order) on 3,000
    random neighbourhoods of 0..12 agents."""
    lib = load_oracle("f64")
    rng = np.random.default_rng(7)
    out = np.zeros(2)
    n_forced = 0
    for k in range(3000):
        params = np.array([rng.uniform(0.05, 3.0), 1.0, 0.0, rng.uniform(0.1, 2.0), rng.uniform(0.5, 3.0),
                           rng.uniform(0.1, 0.6)])
        n = int(rng.integers(0, 13))
        ids = rng.permutation(64)[:n + 1]
        me = _random_record(rng, ids[0], True, 1.0)
        others = np.array([_random_record(rng, ids[1 + q], k % 3 == 0, 1.5) for q in range(n)]).reshape(n, 7)
        rec = rng.normal(0.0, 1.3, 2)
        buf = np.ascontiguousarray(others if n else np.zeros((1, 7)))
        t_o = lib.oracle_zanlungo_desired_velocity(_ptr(params), _ptr(me), _ptr(buf), n, rec[0], rec[1], _ptr(out))
        z = zr.Zanlungo(*params)
        a = zr.Agent(me[0], me[1:3], me[3:5], me[5:7])
        nb = [zr.Agent(o[0], o[1:3], o[3:5], o[5:7]) for o in others]
        with np.errstate(all="ignore"):
            t_r = z.compute_tti(a, nb)
            v = z.get_desired_velocity(a, nb, rec)
        assert _same(t_o, t_r), (k, t_o, t_r)
        assert _same(out[0], v[0], 1e-11) and _same(out[1], v[1], 1e-11), (k, out, v)
        n_forced += int(np.isfinite(t_r))
    assert n_forced > 1000


def test_device_force_formula_equals_the_restatement_for_the_live_branch():
    """What the kernel implements (cs_device_types.hip.inc zanlungo_forward_force) is the folded
    form of the stationary-neighbour branch: F = perp/|perp| * min(1e15, 2A|v_i|/t) *
    exp(-(|p_i + v_i t - p_j| - 2R)/D), perp = +-(-(p_i-p_j).y, (p_i-p_j).x) on the side of v_i.
    Checked here against the restatement in f64, so the folding (slerp(1, d, perp, s) -> perp,
    normalize() removing sin(asin s)/s) is pinned independently of the kernel's f32 rounding."""
    rng = np.random.default_rng(99)
    worst = 0.0
    for k in range(4000):
        A, D, m, R = rng.uniform(0.1, 2.0), rng.uniform(0.1, 1.0), 2.0, rng.uniform(0.05, 0.4)
        z = zr.Zanlungo(A, 1, 0, D, m, R)
        pi, pj = rng.uniform(-2, 2, 2), rng.uniform(-2, 2, 2)
        vi, vj = rng.normal(0, 1.3, 2), rng.normal(0, 1.3, 2)
        t = rng.uniform(0.01, 5.0)
        f = z.compute_agent_force(zr.Agent(3, pi, vi, rng.normal(0, 1, 2)), zr.Agent(9, pj, vj, (0, 0)), t)
        q = pi - pj
        perp = np.array([-q[1], q[0]])
        if perp @ vi < 0:
            perp = -perp
        d = (pi + vi * t) - pj
        mag = min(1e15, 2.0 * A * np.hypot(*vi) / t) * np.exp(-(np.hypot(*d) - 2 * R) / D)
        s = abs(perp[0] * d[1] - perp[1] * d[0])
        ref = perp / np.hypot(*perp) * mag if s > 0 else np.array([np.nan, np.nan])
        err = np.hypot(f[0] - ref[0], f[1] - ref[1]) / max(np.hypot(*ref), 1e-300)
        worst = max(worst, err)
    assert worst < 1e-9, worst
