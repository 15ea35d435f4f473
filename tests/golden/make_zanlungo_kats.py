#!/usr/bin/env python3
"""Hand-derived known answers for the Zanlungo force (zanlungo.rs:23-28, 93-217), written out as data.

The reference holds NO vector for compute_agent_force / right_of_way_vel / slerp, so the only
evidence independent of the two restatements (oracle/crowdstep_oracle.cpp, oracle/zanlungo_restatement.py) is arithmetic
done by hand from the Rust.  This file is that arithmetic in executable form: every case is a geometry chosen so that
each intermediate (roots of the quadratic, future positions, the perpendicular, the cross product) is a short decimal
worked out in the comment beside it, and the answer is ONE closed-form expression in sqrt / exp typed in here.  It
imports neither the oracle nor the restatement nor the engine.  tests/test_zanlungo_kats.py checks the table on the
oracle (pair probe and whole steps), on the numpy restatement and, where Simulation::step can reach the case, on the
device.  DESIGN.md section 6 carries the derivations in prose.

    python tests/golden/make_zanlungo_kats.py      # rewrites tests/golden/zanlungo_kats.json

Conventions: planner Zanlungo(A, obstacle_scale, reaction_time, D, m, R); an agent is (id, position, velocity,
preferred_vel); "scene" cases give the agents BEFORE the step (their velocities are what an earlier step left) and
expect `velocity` = recommended + F / m of the named agent after one step (zanlungo.rs:201-217), with
recommended = that agent's preferred_vel.  In a scene every NEIGHBOUR's preferred_vel is (0, 0) as Simulation::step
hands it over (lib.rs:140,261,271); "pair" cases set it freely and are reachable through the planner alone.
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))
NAN = float("nan")
PLANNER = [1.0, 1.0, 0.0, 1.0, 1.0, 0.5]  # A = 1, D = 1, m = 1, R = 0.5 (SURVEY's KAT-Z1 planner)


def cases():
    out = []

    # ---- Z4: row > 0, own preferred velocity != own velocity -----------------------------------------------------
    # i: id 1, p (0,0), v (1,0), preferred (0,1).  j: id 0, p (3,0), v (0,0).
    # time to collision (both ways): rv = (-1,0), rp = (3,0): a = 1, b = -6, c = 9 - 0.25 = 8.75, disc = 36 - 35 = 1,
    #   t0 = (6 - 1)/2 = 2.5, t1 = 3.5 -> t_i = 2.5
    # i has the LARGER id: right_of_way = clamp(1 - 0) = 1 > 0: r2 = 1, my_vel = v + (pref - v) * 1 = (0,1), other = (0,0),
    #   weight = 1 - 1 = 0.  fut = (0, 2.5), ofut = (3, 0), d = (-3, 2.5), dist = sqrt(15.25); no slerp (weight <= 1);
    #   magnitude = 0 * 1 * |(0,1)| / 2.5 = 0 -> F = (d / |d|) * (0 * exp(..)) = (-0, +0)
    #   velocity = preferred + F/m = (0 + -0, 1 + 0) = (0, 1)
    out.append({
        "name": "Z4", "kind": "scene", "what": "row > 0 (larger id): my_vel = own PREFERRED velocity, weight 0, term exactly (-0, +0)",
        "rust": "zanlungo.rs:190-194,107,159-169", "planner": PLANNER, "grid": [40.0, 40.0, 20.0, -20.0, -20.0], "eyesight": 10.0,
        "agents": [{"id": 0, "p": [3.0, 0.0], "v": [0.0, 0.0], "pref": [0.0, 0.0]},
                   {"id": 1, "p": [0.0, 0.0], "v": [1.0, 0.0], "pref": [0.0, 1.0]}],
        "expect": {"1": {"t_i": 2.5, "velocity": [0.0, 1.0]}, "0": {"t_i": 2.5, "velocity": [0.0, 0.0]}},
        "device": True,
    })

    # ---- Z5: moving neighbour, interpolate (zanlungo.rs:126-135) --------------------------------------------------
    # i: id 0, p (0,0), v (1,0).  j: id 1, p (0,-2), v (0,0), preferred (0.6, 0.8) (|.| = 1 >= 1e-4: the moving branch).
    # t_i = 2 is GIVEN (pair case).  right_of_way = -1: r2 = 1, other_adjusted = v_j + (pref_j - v_j) = (0.6,0.8), weight 2.
    # fut = (2,0); ofut = (0,-2) + 2 (0.6,0.8) = (1.2,-0.4); d = (0.8, 0.4); dist = sqrt(0.8)
    # pref_dir . d = 0.48 + 0.32 = 0.8 > 0: perp = (-0.8, 0.6); perp . d = -0.64 + 0.24 = -0.4 < 0 -> perp = (0.8,-0.6)
    # sin_theta = perp.x d.y - perp.y d.x = 0.32 + 0.48 = 0.8; slerp(1, d, perp, 0.8) = d sin(0)/0.8 + perp sin(asin 0.8)/0.8 = perp
    # normalize -> (0.8,-0.6); magnitude = 2 * 1 * |(1,0) - (0.6,0.8)| / 2 = sqrt(0.16 + 0.64) = sqrt(0.8)
    # F = (0.8,-0.6) * sqrt(0.8) * exp(-(sqrt(0.8) - 1))
    m = math.sqrt(0.8) * math.exp(-(math.sqrt(0.8) - 1.0))
    out.append({
        "name": "Z5", "kind": "pair", "what": "moving neighbour heading into d_ij: perpendicular to ITS preferred velocity, slerp to it",
        "rust": "zanlungo.rs:126-135,142-150", "planner": PLANNER, "t_i": 2.0,
        "me": {"id": 0, "p": [0.0, 0.0], "v": [1.0, 0.0], "pref": [1.0, 0.0]},
        "other": {"id": 1, "p": [0.0, -2.0], "v": [0.0, 0.0], "pref": [0.6, 0.8]},
        "expect": {"force": [0.8 * m, -0.6 * m]}, "rel": 4e-15, "device": False,
    })

    # ---- Z6: moving neighbour, NOT interpolating (zanlungo.rs:136-138) --------------------------------------------
    # i: id 0, p (0,0), v (1,0).  j: id 1, p (3,0), v (0,0), preferred (0,1).  t_i = 2.5 given.
    # other_adjusted = (0,1), weight 2.  fut = (2.5,0); ofut = (3, 2.5); d = (-0.5,-2.5); dist = sqrt(6.5)
    # pref_dir . d = -2.5, not > 0 -> interpolate = false: d stays.  magnitude = 2 |(1,0) - (0,1)| / 2.5 = 2 sqrt2 / 2.5
    # F = d/|d| * magnitude * exp(-(sqrt(6.5) - 1))
    r = math.sqrt(6.5)
    m = 2.0 * math.sqrt(2.0) / 2.5 * math.exp(-(r - 1.0))
    out.append({
        "name": "Z6", "kind": "pair", "what": "moving neighbour heading away from d_ij: no interpolation, force along d_ij",
        "rust": "zanlungo.rs:136-138", "planner": PLANNER, "t_i": 2.5,
        "me": {"id": 0, "p": [0.0, 0.0], "v": [1.0, 0.0], "pref": [1.0, 0.0]},
        "other": {"id": 1, "p": [3.0, 0.0], "v": [0.0, 0.0], "pref": [0.0, 1.0]},
        "expect": {"force": [-0.5 / r * m, -2.5 / r * m]}, "rel": 4e-15, "device": False,
    })

    # ---- Z7: sin_theta > 1 is clamped to 1 (zanlungo.rs:146-148) ---------------------------------------------------
    # i: id 0, p (0,0), v (2,0), preferred (2,0).  j: id 1, p (4,0), v (0,0).
    # rv = (-2,0), rp = (4,0): a = 4, b = -16, c = 15.75, disc = 256 - 252 = 4, t0 = (16 - 2)/8 = 1.75, t1 = 2.25 -> t_i = 1.75
    # weight 2, my_vel (2,0), other (0,0).  fut = (3.5,0), ofut = (4,0), d = (-0.5,0), dist 0.5
    # curr_rel_pos = (-4,0): perp = (-0,-4); perp . v = -0, not < 0.  cross = (-0)(0) - (-4)(-0.5) = -2 -> 2 -> clamped to 1
    # slerp(1, d, perp, 1): theta = pi/2: d * sin(0)/1 + perp * sin(pi/2)/1 = (-0,-4) -> normalized (-0,-1)
    # magnitude = 2 * 2 / 1.75; exp(-(0.5 - 1)) = e^0.5.  F = (-0, -(4/1.75) e^0.5); velocity = (2, -(4/1.75) e^0.5)
    f = 4.0 / 1.75 * math.exp(0.5)
    out.append({
        "name": "Z7", "kind": "scene", "what": "|cross| = 2 > 1: sin_theta clamped to 1, theta = pi/2, direction = perp exactly",
        "rust": "zanlungo.rs:142-150,23-28", "planner": PLANNER, "grid": [40.0, 40.0, 20.0, -20.0, -20.0], "eyesight": 10.0,
        "agents": [{"id": 0, "p": [0.0, 0.0], "v": [2.0, 0.0], "pref": [2.0, 0.0]},
                   {"id": 1, "p": [4.0, 0.0], "v": [0.0, 0.0], "pref": [0.0, 0.0]}],
        "expect": {"0": {"t_i": 1.75, "velocity": [2.0, -f]}, "1": {"t_i": 1.75, "velocity": [0.0, 0.0]}},
        "rel": 4e-15, "device": True,
    })

    # ---- Z7b: sin_theta = 0.6 < 1, no clamp ----------------------------------------------------------------------
    # i: id 0, p (0,0), v (1,0).  j: id 1, p (1.2,0), v (0,0).
    # rv = (-1,0), rp = (1.2,0): a = 1, b = -2.4, c = 1.44 - 0.25 = 1.19, disc = 5.76 - 4.76 = 1, t0 = 0.7, t1 = 1.7 -> 0.7
    # fut = (0.7,0), d = (-0.5,0), dist 0.5; perp = (-0,-1.2); cross = -(-1.2)(-0.5) = -0.6 -> 0.6 (no clamp)
    # slerp(1, d, perp, 0.6) = d * 0 + perp * (sin(asin 0.6)/0.6): a positive multiple of perp -> normalized (-0,-1)
    # magnitude = 2 * 1 / 0.7; F = (-0, -(2/0.7) e^0.5)
    f = 2.0 / 0.7 * math.exp(0.5)
    out.append({
        "name": "Z7b", "kind": "scene", "what": "|cross| = 0.6: unclamped; normalize() removes sin(asin s)/s, direction = perp / |perp|",
        "rust": "zanlungo.rs:142-150,159", "planner": PLANNER, "grid": [40.0, 40.0, 20.0, -20.0, -20.0], "eyesight": 10.0,
        "agents": [{"id": 0, "p": [0.0, 0.0], "v": [1.0, 0.0], "pref": [1.0, 0.0]},
                   {"id": 1, "p": [1.2, 0.0], "v": [0.0, 0.0], "pref": [0.0, 0.0]}],
        "expect": {"0": {"t_i": 0.7, "velocity": [1.0, -f]}, "1": {"t_i": 0.7, "velocity": [0.0, 0.0]}},
        "rel": 1e-14, "t_rel": 1e-15, "device": True,
    })

    # ---- Z8a: s = 0 with d != 0 (d parallel to perp) -> 0/0 in slerp -> NaN ----------------------------------------
    # i: id 0, p (0,0), v (1,1).  j: id 1, p (0,2), v 0.  k: id 2, p (2.5,2), v 0.
    # i-k: rv = (-1,-1), rp = (2.5,2): a = 2, b = 2(-2.5 - 2) = -9, c = 6.25 + 4 - 0.25 = 10, disc = 81 - 80 = 1, t0 = 8/4 = 2, t1 = 2.5
    # i-j: rp = (0,2): a = 2, b = -4, c = 3.75, disc = 16 - 30 < 0 -> +inf.  t_i = 2.
    # pair (i,j): fut = (2,2), ofut = (0,2), d = (2,0); curr_rel_pos = (0,-2): perp = (2,0); perp . v = 2 > 0
    #   cross = 2*0 - 0*2 = 0 -> sin_theta = 0: theta = 0, t0 = sin(0)/0 = NaN -> F = NaN; the sum is NaN
    out.append({
        "name": "Z8a", "kind": "scene", "what": "s = 0 (predicted separation parallel to perp): slerp divides 0 by 0, velocity NaN",
        "rust": "zanlungo.rs:23-28,142-150", "planner": PLANNER, "grid": [40.0, 40.0, 20.0, -20.0, -20.0], "eyesight": 10.0,
        "agents": [{"id": 0, "p": [0.0, 0.0], "v": [1.0, 1.0], "pref": [1.0, 1.0]},
                   {"id": 1, "p": [0.0, 2.0], "v": [0.0, 0.0], "pref": [0.0, 0.0]},
                   {"id": 2, "p": [2.5, 2.0], "v": [0.0, 0.0], "pref": [0.0, 0.0]}],
        "expect": {"0": {"t_i": 2.0, "velocity": [NAN, NAN]}},
        "device": True,
    })

    # ---- Z8b: d = 0 exactly (s = 0 too) ------------------------------------------------------------------------------
    # i: id 0, p (0,0), v (1,0).  j: id 1, p (1,0), v (1,0) (same velocity: a = 0 -> -b/0 ... NaN comparisons -> +inf).
    # k: id 2, p (0,2.5), v (1,-2): rv = (0,-2), rp = (0,2.5): a = 4, b = 2(-5) = -10, c = 6.25 - 0.25 = 6,
    #   disc = 100 - 96 = 4, t0 = (10 - 2)/8 = 1, t1 = 1.5 -> t_i = 1 (every intermediate is exact in f32 as well)
    # pair (i,j): other_adjusted = v_j + (0 - v_j) = (0,0); fut = (1,0) = ofut: d = (0,0), dist 0; perp = (-0,-1); cross = 0 -> NaN
    out.append({
        "name": "Z8b", "kind": "scene", "what": "predicted positions coincide (the neighbour's velocity is blended to its zero preferred velocity): NaN",
        "rust": "zanlungo.rs:185-189,109-112,23-28", "planner": PLANNER, "grid": [40.0, 40.0, 20.0, -20.0, -20.0], "eyesight": 10.0,
        "agents": [{"id": 0, "p": [0.0, 0.0], "v": [1.0, 0.0], "pref": [1.0, 0.0]},
                   {"id": 1, "p": [1.0, 0.0], "v": [1.0, 0.0], "pref": [1.0, 0.0]},
                   {"id": 2, "p": [0.0, 2.5], "v": [1.0, -2.0], "pref": [1.0, -2.0]}],
        "expect": {"0": {"t_i": 1.0, "velocity": [NAN, NAN]}},
        "device": True,
    })

    # ---- Z9: three terms, summed in canonical order (cells x-major / y-minor, not id order) ---------------------------
    # grid 20 x 20, cell 1, offset (-10,-10).  i: id 0, p (0,0), v (1,0).
    # j: id 1, p (3,0) cell (13,10);  k: id 2, p (3,1) cell (13,11);  l: id 3, p (3,-1.5) cell (13,8).  All stationary.
    # t_i: j as in KAT-Z1 -> 2.5; k: c = 9 + 1 - 0.25 = 9.75, disc = 36 - 39 < 0; l: c = 11, disc = 36 - 44 < 0.  t_i = 2.5, fut = (2.5, 0)
    # every term: weight 2, magnitude 2/2.5 = 0.8, |cross| >= 1 -> direction = perp / |perp|
    #  j: d = (-0.5, 0), dist 0.5;  perp = (-0,-3)           -> (-0,-1)        * 0.8 * e^(0.5)
    #  k: d = (-0.5,-1), dist sqrt(1.25); curr = (-3,-1): perp = (1,-3), perp . v = 1 > 0; cross = -1 - 1.5 = -2.5
    #     -> (1,-3)/sqrt10 * 0.8 * exp(-(sqrt(1.25) - 1))
    #  l: d = (-0.5, 1.5), dist sqrt(2.5); curr = (-3, 1.5): perp = (-1.5,-3), perp . v = -1.5 < 0 -> (1.5, 3); cross = 1.5*1.5 - 3*(-0.5) = 3.75
    #     -> (1.5,3)/sqrt(11.25) * 0.8 * exp(-(sqrt(2.5) - 1))
    # canonical order of the neighbour list: cell (13,8) l, then (13,10) j, then (13,11) k
    fj = (-0.0 * 0.8 * math.exp(0.5), -1.0 * 0.8 * math.exp(0.5))
    mk = 0.8 * math.exp(-(math.sqrt(1.25) - 1.0))
    fk = (1.0 / math.sqrt(10.0) * mk, -3.0 / math.sqrt(10.0) * mk)
    ml = 0.8 * math.exp(-(math.sqrt(2.5) - 1.0))
    fl = (1.5 / math.sqrt(11.25) * ml, 3.0 / math.sqrt(11.25) * ml)
    fx = (fl[0] + fj[0]) + fk[0]
    fy = (fl[1] + fj[1]) + fk[1]
    out.append({
        "name": "Z9", "kind": "scene", "what": "three forward neighbours in three cells: the terms are summed l, j, k (cell order), not 1, 2, 3",
        "rust": "zanlungo.rs:211-214; location_hash_2d.rs:245-256", "planner": PLANNER, "grid": [20.0, 20.0, 1.0, -10.0, -10.0], "eyesight": 5.0,
        "agents": [{"id": 0, "p": [0.0, 0.0], "v": [1.0, 0.0], "pref": [1.0, 0.0]},
                   {"id": 1, "p": [3.0, 0.0], "v": [0.0, 0.0], "pref": [0.0, 0.0]},
                   {"id": 2, "p": [3.0, 1.0], "v": [0.0, 0.0], "pref": [0.0, 0.0]},
                   {"id": 3, "p": [3.0, -1.5], "v": [0.0, 0.0], "pref": [0.0, 0.0]}],
        "expect": {"0": {"t_i": 2.5, "velocity": [1.0 + fx, fy], "terms_in_order": [list(fl), list(fj), list(fk)]}},
        "rel": 1e-14, "device": True,
    })
    return out


def main():
    def enc(o):
        if isinstance(o, float) and math.isnan(o):
            return "NaN"
        if isinstance(o, dict):
            return {k: enc(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [enc(v) for v in o]
        return o
    table = {"about": "hand-derived known answers for zanlungo.rs:23-28,93-217; derivations: tests/golden/make_zanlungo_kats.py and "
                      "DESIGN.md section 6; NOT outputs of the oracle, the restatement or the engine",
             "agent": "id, p = position, v = velocity, pref = preferred_vel (lib.rs:46-65)",
             "cases": enc(cases())}
    with open(os.path.join(HERE, "zanlungo_kats.json"), "w") as f:
        json.dump(table, f, indent=1)
    print("wrote", len(table["cases"]), "cases")


if __name__ == "__main__":
    main()
