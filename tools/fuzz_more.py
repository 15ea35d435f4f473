#!/usr/bin/env python3
"""Further seeds of the randomized mesh tests (tests/test_gpu_tiles.py), by hand: python tools/fuzz_more.py FIRST LAST
Each seed runs with the usual launches and with the border / interior launches (CS_TILE_SPLIT=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402
import test_gpu_tiles as T  # noqa: E402


class Env:
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


first, last = int(sys.argv[1]), int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "mesh"
if os.environ.get("CS_FUZZ_NATIVE") == "1":
    # the same randomized mesh tests with the C ABI's mesh (cs_mesh_*, tiles.NativeTileMesh) standing in for the
    # Python orchestration (tiles.LocalTileMesh)
    from rmf_crowdsim_amd.tiles import NativeTileMesh

    class NativeAsLocal(NativeTileMesh):
        def __init__(self, index, tiles, halo_cells=1, phases=1, **kw):
            if phases != 1:
                raise pytest.skip.Exception("the native mesh has the one-phase schedule only")
            super().__init__(index, tiles, halo_cells, **kw)

        @property
        def engines(self):  # (the tests sum reports and counts over a mesh's engines: the mesh's own are those sums)
            return [self]
    T.LocalTileMesh = NativeAsLocal
if which == "bigmesh":  # crowds large enough for the LDS-tiled kernel on every tile: border / interior launches, fused pack
    import math
    import numpy as np
    from rmf_crowdsim_amd import LocationHash2D, Simulation, StubHighLevelPlan, Zanlungo, scenes
    from rmf_crowdsim_amd.tiles import LocalTileMesh
    ran = 0
    for seed in range(first, last):
        rng = np.random.default_rng(77000 + seed)
        cell = float(rng.choice([1.0, 1.5, 2.0, 3.0, 4.0])); eyes = float(rng.choice([1.0, 2.0, 3.0]))
        halo = max(1, math.ceil(eyes / cell - 1e-9))
        n = int(rng.integers(30000, 90000))
        pts, grid, extent, group = scenes.uniform_crowd(n, seed=seed, cell_size=cell, margin=float(rng.choice([2.0 * cell, 10.0, 30.0])))
        tiles = [(2, 2), (3, 1), (1, 3), (2, 3), (4, 2), (1, 2)][int(rng.integers(0, 6))]
        ncell = int(grid["width"] / cell)
        if min(ncell // tiles[0], ncell // tiles[1]) < 2 * halo + 2: tiles = (2, 1)
        walk = float(rng.choice([0.002, 0.3, 1.3])); lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
        vel = [(walk, 0.3 * walk + 0.001), (walk * 0.98, 0.3 * walk - 0.001)]
        outs = []
        for kind in ("single", "mesh", "split"):
            os.environ["CS_TILE_SPLIT"] = "1" if kind == "split" else "0"
            t = Simulation(LocationHash2D(**grid)) if kind == "single" else LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=halo)
            for g, v in enumerate(vel):
                t.add_agents(pts[group == g], StubHighLevelPlan(v), lp, eyes)
            for k in range(12):
                t.step(0.05, report=(k == 5))
                if k == 6:
                    # (in the empty margin, beside the middle of the grid: an agent dropped INTO the lattice overlaps a
                    # walker, both go NaN (KAT-Z3) and are binned to cell 0, which the split launches refuse: DESIGN.md section 7)
                    extra = np.array([[grid["offset"][0] + 1.0, grid["offset"][1] + grid["height"] / 2 + 0.3]])
                    t.add_agents(extra, StubHighLevelPlan((0.1, 0.1)), lp, eyes)
            outs.append(t.read_agents())
            del t
        assert outs[0].tobytes() == outs[1].tobytes() == outs[2].tobytes(), f"seed {seed}: cell {cell} eyes {eyes} tiles {tiles} n {n}"
        ran += 1
        print(f"seed {seed} ok (cell {cell}, eyesight {eyes}, tiles {tiles}, {n} agents, walk {walk})", flush=True)
    print(f"{ran} cases passed")
    sys.exit(0)
if which == "kernels":  # the LDS-tiled kernel against the gather kernel, bit for bit, on crowds of every texture
    import numpy as np
    from rmf_crowdsim_amd import LocationHash2D, Simulation, StubHighLevelPlan, Zanlungo, scenes
    ran = 0
    for seed in range(first, last):
        rng = np.random.default_rng(99000 + seed)
        cell = float(rng.choice([1.0, 1.5, 2.0, 3.0, 4.0]))
        eyes = float(rng.choice([0.8, 1.0, 2.0, 3.0]))
        n = int(rng.integers(20000, 70000))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            pts, grid, extent, group = scenes.uniform_crowd(n, seed=seed, cell_size=cell)
        elif kind == 1:
            pts, grid, extent, group = scenes.random_crowd(n, seed=seed, cell_size=cell)
        else:
            pts, grid, extent, group = scenes.hotspot_crowd(n, seed=seed, cell_size=cell, per_hotspot=int(rng.choice([300, 800, 2000])))
        # ids uncorrelated with the groups: four add_agents calls over a random split
        part = rng.integers(0, 4, size=len(pts))
        speed = float(rng.choice([1e-3, 1e-2, 0.3]))
        dense = kind == 2
        outs = []
        for flags in (2, 1):
            sim = Simulation(LocationHash2D(**grid), flags=flags | (4 if dense else 0))
            lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
            for k in range(4):
                v = (speed * (1 if k % 2 else -1), speed * 0.3 * (k - 1.5))
                sim.add_agents(pts[part == k], StubHighLevelPlan(v), lp, eyes)
            try:
                for _ in range(int(os.environ.get("CS_FUZZ_STEPS", "6"))):
                    sim.step(0.05, report=False)
                outs.append(sim.read_agents().tobytes())
            except Exception as err:
                outs.append("error: " + str(err))
            del sim
        assert outs[0] == outs[1], f"seed {seed}: kind {kind} cell {cell} eyes {eyes} n {n} speed {speed}"
        ran += 1
        print(f"seed {seed} ok (kind {kind}, cell {cell}, eyesight {eyes}, {n} agents, speed {speed}"
              f"{', both fail alike: ' + outs[0] if isinstance(outs[0], str) else ''})", flush=True)
    print(f"{ran} cases passed")
    sys.exit(0)
if which == "parity":  # the engine against the oracle (tests/test_gpu_parity.py), further seeds
    import test_gpu_parity as P
    ran = 0
    stops = []  # (CS_FUZZ_KEEP_GOING=1: note a failing case and go on, for a look at many seeds)
    for seed in range(first, last):
        for fn in (P.test_random_configurations_match_oracle_and_each_other,
                   P.test_random_removals_and_queries_between_steps_match_the_oracle,
                   P.test_random_planner_groups_match_oracle_and_each_other,
                   P.test_random_api_sequences_give_the_oracle_s_results_and_errors):
            try:
                fn(seed)
                ran += 1
            except pytest.skip.Exception:
                pass
            except P.CrowdSimError as err:
                # a scene that leaves the model's finite range (the test does not expect one): fine if the
                # ORACLE leaves it too -- the same test with the oracle standing in for the engine
                engine = P.Simulation
                P.Simulation = lambda index, flags=0, **kw: P.OracleSimulation(index)
                try:
                    fn(seed)
                    print(f"FAILED {fn.__name__} seed {seed}: the engine raised '{err}', the oracle did not", flush=True)
                    raise err
                except P.CrowdSimError as err2:
                    print(f"  seed {seed}: {fn.__name__}: engine '{err}', oracle '{err2}' (the model left its range)")
                except AssertionError:
                    # (oracle against oracle can trip the test's own checks before it gets to the failing step: the
                    # oracle's k-NN is the reference's inexact ring search, the test compares with brute force)
                    print(f"  seed {seed}: {fn.__name__}: engine '{err}'; the oracle-only rerun stopped at an assertion of the "
                          f"test itself: replay the seed on the oracle by hand", flush=True)
                    stops.append((fn.__name__, seed, "engine raised; oracle-only rerun inconclusive"))
                finally:
                    P.Simulation = engine
            except Exception as err:
                print(f"FAILED {fn.__name__} seed {seed}: {str(err)[:160]}", flush=True)
                if os.environ.get("CS_FUZZ_KEEP_GOING") != "1":
                    raise
                why = ""
                if fn is P.test_random_api_sequences_give_the_oracle_s_results_and_errors:
                    # triage against BOTH builds of the oracle: where the model has thrown agents (overlaps, the
                    # 1e15 clamp) the f64 and the f32 reading of the reference part ways themselves
                    import numpy as np

                    class O32(P.OracleSimulation):
                        _kind = "f32"

                    def digest(log):
                        out = []
                        for c in log:
                            if c[0] == "final" and c[1] == "ok":
                                ids, xs = np.array(c[2][0]), np.array(c[2][1], dtype=float)
                                out.append(("final", len(ids), tuple(int(i) for i in ids[np.isnan(xs)])))
                            else:
                                out.append((c[0], c[1], c[2] if c[1] == "err" else None))
                        return out
                    de, d64, d32 = (digest(P._api_sequence(cls, 12000 + seed)) for cls in (P.Simulation, P.OracleSimulation, O32))
                    why = ("statuses and NaN sets equal: values only" if de == d64 else
                           "engine == f32 oracle != f64 oracle" if de == d32 else
                           "f64 == f32 oracle != engine" if d64 == d32 else "all three differ")
                    print(f"  triage: {why}", flush=True)
                stops.append((fn.__name__, seed, why))
        print(f"seed {seed} ok", flush=True)
    print(f"{ran} cases passed; stops: {stops}")
    sys.exit(0)
if which == "sinks":  # the source-sink / route-follower scenes (engine, oracle and a mesh), with the env set for the mesh
    ran = 0
    for seed in range(first, last):
        for split in ("0", "1"):
            os.environ["CS_TILE_SPLIT"] = split
            for fn in (T.test_random_source_sinks_engine_oracle_and_mesh_agree, T.test_random_route_followers_engine_oracle_and_mesh_agree):
                try:
                    fn(seed)
                    ran += 1
                except pytest.skip.Exception:
                    pass
                except AssertionError as err:
                    if "[None, None, (" in str(err) and "Index out of bounds" in str(err):
                        # the mesh alone failed: a tile's grid edges are strict, the single engine clamps below the
                        # grid and ALIASES above the row stride like the reference
                        # (seed 177: a walker leaving through y = 80)
                        print(f"  seed {seed}: only the mesh failed with 'Index out of bounds' ({fn.__name__}): strict tile edges?")
                        continue
                    if str(err).startswith("[None, (") and str(err).rstrip().endswith("None]"):
                        # the f64 oracle alone left the grid: the reference's own f64 artifacts (|rel_vel|^2 underflow
                        # reads as "colliding now"), which f32 cannot reproduce
                        print(f"  seed {seed}: only the f64 oracle failed ({fn.__name__}): the reference's f64 artifact?")
                        continue
                    if split == "1" and "halo band" in str(err):  # a NaN agent (the model blew up) binned to cell 0:
                        print(f"  seed {seed}: the split launches refuse a NaN agent ({fn.__name__})")  # documented limit
                        continue
                    print(f"FAILED {fn.__name__} seed {seed} split {split}", flush=True)
                    raise
                except Exception:
                    print(f"FAILED {fn.__name__} seed {seed} split {split}", flush=True)
                    raise
            for tiles, phases in (((4, 2), 1), ((2, 2), 1), ((1, 3), 1), ((1, 2), 1)):
                try:
                    T.test_creeping_crowd_with_random_source_sinks_across_tiles(seed, tiles, phases)
                    ran += 1
                except Exception:
                    print(f"FAILED creeping crowd seed {seed} tiles {tiles} split {split}", flush=True)
                    raise
        print(f"seed {seed} ok", flush=True)
    print(f"{ran} cases passed")
    sys.exit(0)
ran = skipped = 0
for seed in range(first, last):
    for split in (False, True):
        for fn in (T.test_random_meshes_match_the_single_engine, T.test_random_call_sequences_on_a_mesh_match_the_single_engine):
            os.environ.pop("CS_TILE_SPLIT", None)
            try:
                fn(seed, split, Env())
                ran += 1
            except pytest.skip.Exception:
                skipped += 1
            except Exception:
                print(f"FAILED {fn.__name__} seed {seed} split {split}", flush=True)
                raise
    print(f"seed {seed} ok", flush=True)
print(f"{ran} cases passed, {skipped} skipped")
