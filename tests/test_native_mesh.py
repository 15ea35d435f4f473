"""The multi-tile crowd behind the C ABI's mesh handle (cs_mesh_*, csrc/cs_mesh.hip.inc; review of round 2:
layout, capacities, re-cuts, route-miss and query merging lived only in Python).  NativeTileMesh is a thin binding;
what it drives must equal the single engine bit for bit, like the Python orchestration it replaces (LocalTileMesh,
tests/test_gpu_tiles.py, whose scenes are reused here).

CPU: the binding against the ORACLE's cs_mesh_* (one reference simulation whatever the tiling): the plumbing.
"""
import numpy as np
import pytest

from oracle_sim import OracleSimulation, load_oracle
from rmf_crowdsim_amd import (LocationHash2D, NoLocalPlan, Simulation, StubHighLevelPlan, Zanlungo, scenes)
from rmf_crowdsim_amd.tiles import LocalTileMesh, NativeTileMesh
from test_gpu_tiles import _multi_leg_scene, _sink_scene


class Heard:
    def __init__(self):
        self.added, self.removed = [], []

    def agent_spawned(self, position, agent):
        self.added.append(agent)

    def agent_destroyed(self, agent):
        self.removed.append(agent)


def test_binding_against_the_oracles_mesh():
    grid = dict(width=60.0, height=60.0, cell_size=2.0, offset=(0.0, 0.0))
    mesh = NativeTileMesh(LocationHash2D(**grid), (2, 2), 1, library=load_oracle("f64"))
    ora = OracleSimulation(LocationHash2D(**grid))
    heard = Heard()
    mesh.add_event_listener(heard)
    for t in (mesh, ora):
        _sink_scene(t)
        t.add_agents([(30.0, 30.0), (31.0, 30.5)], StubHighLevelPlan((0.1, 0.0)), NoLocalPlan(), 2.0)
    for k in range(120):
        mesh.step(0.05)
        ora.step(0.05)
    mesh.remove_agents(1)
    ora.remove_agents(1)
    a, b = mesh.read_agents(), ora.read_agents()
    assert len(a) > 20 and a.tobytes() == b.tobytes() and len(mesh) == len(ora)
    assert heard.added[:2] == [0, 1] and 1 in heard.removed and len(heard.added) == int(a["id"].max()) + 1
    assert mesh.get_neighbours_in_radius(3.0, (30.0, 30.0)) == ora.get_neighbours_in_radius(3.0, (30.0, 30.0))
    assert mesh.get_nearest_neighbours(2, (30.0, 30.0)) == ora.get_nearest_neighbours(2, (30.0, 30.0))
    with pytest.raises(Exception):
        NativeTileMesh(LocationHash2D(**grid), (0, 2), 1, library=load_oracle("f64"))


@pytest.mark.gpu
@pytest.mark.parametrize("tiles,halo,cell", [((2, 2), 1, 2.0), ((3, 1), 1, 2.0), ((2, 3), 2, 1.0)])
def test_creeping_crowd_on_a_native_mesh_matches_engine_and_python_mesh(tiles, halo, cell):
    pts, grid, extent, group = scenes.uniform_crowd(40000, seed=13, cell_size=cell)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    runs = []
    for make in (lambda: Simulation(LocationHash2D(**grid)),
                 lambda: NativeTileMesh(LocationHash2D(**grid), tiles, halo, density_per_cell=4.0 * cell * cell),
                 lambda: LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=halo, density_per_cell=4.0 * cell * cell)):
        t = make()
        scenes.add_walking_crowd(t, pts, group, lp, 2.0)   # 6.5 cm per step: agents migrate over the cuts
        for k in range(40):
            t.step(0.05, report=(k % 16 == 0))
        runs.append(t.read_agents())
    assert len(runs[0]) == 40000 and runs[0].tobytes() == runs[1].tobytes() == runs[2].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("tiles,listen", [((2, 2), False), ((3, 1), True)])
def test_source_sinks_on_a_native_mesh(tiles, listen):
    """Spawn flags OR-ed over the tiles in the library: on the device while nobody listens, through the host with a
    listener or a report; ids, events and positions as the single engine's."""
    grid = dict(width=60.0, height=60.0, cell_size=2.0, offset=(0.0, 0.0))
    single, mesh = Simulation(LocationHash2D(**grid)), NativeTileMesh(LocationHash2D(**grid), tiles, 1)
    heard_s, heard_m = Heard(), Heard()
    if listen:
        single.add_event_listener(heard_s)
        mesh.add_event_listener(heard_m)
    for t in (single, mesh):
        _sink_scene(t)
    for k in range(800):
        with_report = k in (250, 251, 500)
        single.step(0.05, report=with_report)
        mesh.step(0.05, report=with_report)
        if with_report:
            assert {key: single.last_report[key] for key in ("n_agents", "n_spawned", "n_destroyed")} == \
                   {key: mesh.last_report[key] for key in ("n_agents", "n_spawned", "n_destroyed")}
    a, b = single.read_agents(), mesh.read_agents()
    assert len(a) > 100 and a["id"].max() > 300 and a.tobytes() == b.tobytes()
    if listen:  # every spawn and removal heard once (per-tile order: sort)
        assert sorted(heard_s.added) == sorted(heard_m.added) and sorted(heard_s.removed) == sorted(heard_m.removed)
        assert len(heard_m.removed) > 100
    victim = int(a["id"][5])
    single.remove_agents(victim)
    mesh.remove_agents(victim)
    with pytest.raises(Exception, match="unknown agent id"):
        mesh.remove_agents(10 ** 9)
    for _ in range(20):
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    assert single.read_agents().tobytes() == mesh.read_agents().tobytes()


@pytest.mark.gpu
def test_multi_leg_route_followers_on_a_native_mesh():
    """Route-cache misses of all tiles merged in agent order and planned by every tile, inside cs_mesh_step."""
    grid = dict(width=160.0, height=160.0, cell_size=2.0, offset=(0.0, 0.0))
    single, mesh = Simulation(LocationHash2D(**grid)), NativeTileMesh(LocationHash2D(**grid), (2, 2), 1)
    r_single, r_mesh = (_multi_leg_scene(t, NoLocalPlan()) for t in (single, mesh))
    for k in range(700):
        single.step(0.1, report=False)
        mesh.step(0.1, report=False)
    a, b = single.read_agents(), mesh.read_agents()
    assert len(a) > 100 and (a["next_waypoint"] == 1).sum() > 20 and a.tobytes() == b.tobytes()
    assert len(r_mesh.calls) == 4 * len(r_single.calls) >= 4 * 32   # every tile planned the same routes


@pytest.mark.gpu
def test_recut_and_queries_on_a_native_mesh():
    """configs[4] in miniature: weighted cuts at creation, a re-cut of the running crowd in the library, merged
    radius and k-NN queries; same bits and same lists as the single engine throughout."""
    from rmf_crowdsim_amd import _abi
    n = 120000
    pts, grid, extent, group = scenes.hotspot_crowd(n, seed=11, cell_size=2.0, margin=10.0)
    grid = dict(grid, width=grid["width"] + 60.0, height=grid["height"] + 60.0)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    single = Simulation(LocationHash2D(**grid), flags=_abi.CS_CFG_DENSE)
    even = NativeTileMesh(LocationHash2D(**grid), (4, 2), 1, density_per_cell=60.0, flags=_abi.CS_CFG_DENSE)
    cut = NativeTileMesh(LocationHash2D(**grid), (4, 2), 1, density_per_cell=60.0, flags=_abi.CS_CFG_DENSE, weights=pts)
    for t in (single, even, cut):
        scenes.add_walking_crowd(t, pts, group, lp, 2.0, creep=scenes.CREEP_SPEED * 0.1)
        for _ in range(5):
            t.step(0.05, report=False)
    before, weighted = even.tile_counts(), cut.tile_counts()
    assert before.sum() == weighted.sum() == n
    assert before.max() / before.mean() > 1.25 and weighted.max() / weighted.mean() <= 1.2
    ref = single.read_agents()
    assert ref.tobytes() == even.read_agents().tobytes() == cut.read_agents().tobytes()
    after = even.recut()
    print(f"native recut: max/mean {before.max() / before.mean():.2f} -> {after.max() / after.mean():.3f} "
          f"(cuts at creation: {weighted.max() / weighted.mean():.3f})")
    assert after.sum() == n and after.max() / after.mean() <= 1.2
    assert ref.tobytes() == even.read_agents().tobytes()
    rng = np.random.default_rng(8)
    q = rng.uniform(10.0, 10.0 + extent, (100, 2))
    radii = rng.choice([0.5, 2.0, 6.0, 15.0], 100)
    want = single.query_radius_batch(radii, q)
    assert even.get_neighbours_in_radius_batch(radii, q) == want and sum(len(x) for x in want) > 3000
    assert even.get_nearest_neighbours_batch(7, q[:40]) == single.query_knn_batch(7, q[:40])
    for t in (single, even):
        for _ in range(60):
            t.step(0.05, report=False)
    assert single.read_agents().tobytes() == even.read_agents().tobytes()


@pytest.mark.gpu
def test_distributed_form_with_one_rank():
    """The distributed form of the mesh (a tile per rank, halo records over RCCL from the engine, the whole step one
    call: cs_tile_step_rccl) with the one rank a one-GPU box allows: a 1 x 1 mesh on a communicator of one.  Device
    path, host path (a report), source-sinks, a removal, a re-cut, merged queries: the single engine's bits."""
    grid = dict(width=60.0, height=60.0, cell_size=2.0, offset=(0.0, 0.0))
    single = Simulation(LocationHash2D(**grid))
    uid = single.rccl_unique_id()
    mesh = NativeTileMesh(LocationHash2D(**grid), (1, 1), 1, rccl_unique_id=uid, rank=0, n_ranks=1)
    for t in (single, mesh):
        _sink_scene(t)
    for k in range(400):
        with_report = k in (150, 151)
        single.step(0.05, report=with_report)
        mesh.step(0.05, report=with_report)
    a, b = single.read_agents(), mesh.read_agents()
    assert len(a) > 100 and a.tobytes() == b.tobytes()
    single.remove_agents(int(a["id"][3]))
    mesh.remove_agents(int(a["id"][3]))
    for _ in range(10):
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    assert single.read_agents().tobytes() == mesh.read_agents().tobytes()
    # what needs every rank's answer (agents of the whole crowd, re-cut, merged queries) goes through ncclAllGather /
    # ncclAllReduce from the engine when the host brought no transport of its own: here on the communicator of one
    mesh.recut()
    for _ in range(5):
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    assert single.read_agents().tobytes() == mesh.read_agents().tobytes() and len(mesh) == len(single)
    probes = [(30.0, 30.0), (20.0, 41.0)]
    assert mesh.get_neighbours_in_radius_batch([6.0, 9.0], probes) == single.query_radius_batch([6.0, 9.0], probes)
    assert mesh.get_nearest_neighbours_batch(4, probes) == single.query_knn_batch(4, probes)
    with pytest.raises(Exception, match="unknown agent id"):
        mesh.remove_agents(10 ** 9)


# ---- the distributed form over a transport the HOST brings (cs_mesh_host_transport) ----------------------
def _rank_host_transport(rank, world, port, out_path, kind, layout):
    """One rank per tile, two processes sharing the GPU, torch.distributed / gloo as the host's transport: halo
    records, spawn flags, route-cache misses, a re-cut, a removal and merged queries all go through cs_mesh_*."""
    import os
    import pickle
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_tiles import _TWO_RANK_GRID, _two_rank_scene
    from rmf_crowdsim_amd.tiles import TorchHostTransport
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        side, dt = _TWO_RANK_GRID[kind]
        grid = dict(width=side, height=side, cell_size=2.0, offset=(0.0, 0.0))
        mesh = NativeTileMesh(LocationHash2D(**grid), layout, 2, device=0, rank=rank, n_ranks=world,
                              host_transport=TorchHostTransport(dist))
        _two_rank_scene(kind, mesh)
        for k in range(400):
            if k == 120 and kind == "sinks":
                mesh.remove_source_sink(3)
            if k == 260:
                mesh.recut()  # collective: histograms and exported records through the host's allgather
            mesh.step(dt, report=(k in (150, 151)))
            if k == 200:  # the whole crowd is on every rank: everybody removes the same walker
                mesh.remove_agents(int(mesh.read_agents()["id"].max()))
        a = mesh.read_agents()
        probes = [(side * 0.5, side * 0.5), (side * 0.25, side * 0.6), (side * 0.75, side * 0.4)]
        near = mesh.get_neighbours_in_radius_batch([6.0, 9.0, 4.0], probes)
        knn = mesh.get_nearest_neighbours_batch(5, probes)
        counts = [None] * world
        dist.all_gather_object(counts, int(mesh.tile_counts().sum()))
        if rank == 0:
            with open(out_path, "wb") as f:
                pickle.dump((a, near, knn, counts), f)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,layout,port", [("sinks", (2, 1), 29741), ("legs", (1, 2), 29742), ("sinks", (2, 2), 29743)])
def test_two_ranks_over_a_host_transport(tmp_path, kind, layout, port):
    """The distributed cs_mesh over a transport of the host's (here torch.distributed / gloo: two or four ranks sharing
    the one GPU): same bits as one engine after 400 steps with a removed sink, a removed agent and a re-cut on the way;
    read_agents and the batch queries answer for the whole crowd on every rank."""
    import pickle
    import torch.multiprocessing as mp
    from test_gpu_tiles import _TWO_RANK_GRID, _two_rank_scene
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "host_transport.pkl")
    world = layout[0] * layout[1]  # (2 x 2: four ranks, diagonal neighbours exchange corner records)
    procs = [ctx.Process(target=_rank_host_transport, args=(r, world, port, out, kind, layout)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    with open(out, "rb") as f:
        both, near, knn, counts = pickle.load(f)
    side, dt = _TWO_RANK_GRID[kind]
    single = Simulation(LocationHash2D(side, side, 2.0, (0.0, 0.0)))
    _two_rank_scene(kind, single)
    for k in range(400):
        if k == 120 and kind == "sinks":
            single.remove_source_sink(3)
        single.step(dt, report=False)
        if k == 200:
            single.remove_agents(int(single.read_agents()["id"].max()))
    a = single.read_agents()
    assert len(a) > 50 and a.tobytes() == both.tobytes()
    assert sum(counts) == len(a) and min(counts) > 0
    probes = [(side * 0.5, side * 0.5), (side * 0.25, side * 0.6), (side * 0.75, side * 0.4)]
    assert near == single.query_radius_batch([6.0, 9.0, 4.0], probes)
    assert knn == single.query_knn_batch(5, probes)


# ---- the domain's own edges on a mesh (review of round 3: only edges shared with a neighbour tile are strict) ----
def _edge_scene(t):
    """Walkers that leave the domain over its low-x and low-y edges (the reference bins them into row / column 0 and
    keeps their exact position, location_hash_2d.rs:54-66), watched by Zanlungo neighbours that stay inside."""
    lp = Zanlungo(1.0, 1.0, 0.0, 0.4, 2.0, 0.2)
    t.add_agents([(0.5, 0.5), (0.7, 2.6), (0.6, 9.3)], StubHighLevelPlan((-0.4, 0.0)), NoLocalPlan(), 1.0)   # out over x = 0
    t.add_agents([(7.5, 0.4), (10.6, 0.7)], StubHighLevelPlan((0.0, -0.4)), NoLocalPlan(), 1.0)              # out over y = 0
    t.add_agents([(0.9, 0.8)], StubHighLevelPlan((-0.3, -0.3)), lp, 1.5)                                       # over the corner
    t.add_agents([(1.4, 1.1), (1.2, 2.9), (8.1, 1.2), (1.6, 9.0)], StubHighLevelPlan((0.02, 0.01)), lp, 1.5)   # the watchers
    t.add_agents([(-0.3, 4.2), (5.5, -0.2)], StubHighLevelPlan((0.0, 0.0)), lp, 1.5)   # added outside: clamped at once


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", [(2, 2), (3, 1), (1, 3)])
def test_domain_low_edges_clamp_on_a_mesh_like_the_reference(tiles):
    grid = dict(width=12.0, height=12.0, cell_size=1.0, offset=(0.0, 0.0))
    single, ora = Simulation(LocationHash2D(**grid)), OracleSimulation(LocationHash2D(**grid))
    mesh = NativeTileMesh(LocationHash2D(**grid), tiles, 2)
    local = LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=2)
    for t in (single, ora, mesh, local):
        _edge_scene(t)
        for k in range(12):
            t.step(0.5)
    a, o = single.read_agents(), ora.read_agents()
    assert (a["x"] < 0).sum() >= 5 and (a["y"] < 0).sum() >= 4   # they did leave
    assert np.allclose(a["x"], o["x"], atol=2e-6) and np.allclose(a["y"], o["y"], atol=2e-6)
    assert a.tobytes() == mesh.read_agents().tobytes() == local.read_agents().tobytes()
    assert single.last_report["n_clamped"] == ora.last_report["n_clamped"] == mesh.last_report["n_clamped"] > 0
    for probe, r in (((0.1, 0.5), 3.0), ((0.5, 3.0), 2.5), ((8.0, 0.2), 2.0)):
        want = ora.get_neighbours_in_radius(r, probe)
        assert single.get_neighbours_in_radius(r, probe) == want == mesh.get_neighbours_in_radius(r, probe)


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", [(2, 2), (3, 1)])
def test_nan_positioned_agents_on_a_mesh(tiles):
    """location_to_index casts NaN to 0: such an agent sits in row / column 0 for ever and is inert; on a mesh the
    tile that owns that cell holds it (tests/test_gpu_parity.py has the single-engine form)."""
    def run(make):
        sim = make()
        lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
        ids = sim.add_agents([(1.0, 1.0), (float("nan"), 3.0), (1.6, 1.2), (float("nan"), float("nan")), (25.0, 30.0)],
                             StubHighLevelPlan((0.1, 0.05)), lp, 2.0)
        q = sim.get_neighbours_in_radius(3.0, (1.0, 1.0))
        for _ in range(20):
            sim.step(0.05)
        return ids, q, sim.read_agents(), sim.last_report
    grid = dict(width=40.0, height=40.0, cell_size=2.0, offset=(0.0, 0.0))
    (ig, qg, ag, rg) = run(lambda: Simulation(LocationHash2D(**grid)))
    (im, qm, am, rm) = run(lambda: NativeTileMesh(LocationHash2D(**grid), tiles, 1))
    (io, qo, ao, ro) = run(lambda: OracleSimulation(LocationHash2D(**grid)))
    assert ig == im == io and sorted(qg) == sorted(qm) == sorted(qo) == [0, 2]
    assert len(am) == 5 and np.isnan(am["x"][1]) and np.isnan(am["y"][3])
    # (NaN != NaN bytewise-equal still: same bit patterns from the same arithmetic)
    assert ag.tobytes() == am.tobytes()
    ok = [0, 2, 4]
    assert np.allclose(am["x"][ok], ao["x"][ok], atol=1e-6) and np.allclose(am["y"][ok], ao["y"][ok], atol=1e-6)
    assert rg["n_nonfinite"] == rm["n_nonfinite"] == ro["n_nonfinite"] == 2


# ---- a step that fails on one tile stops the whole mesh (advisor, round 3) -------------------------------------
def _runaway_scene(t):
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    t.add_agents([(10.0, 10.0), (30.0, 10.0), (10.0, 30.0), (30.0, 31.0), (20.0, 20.5)], StubHighLevelPlan((0.1, 0.0)), lp, 2.0)
    return t.add_agents([(37.1, 30.0)], StubHighLevelPlan((4.0, 0.0)), NoLocalPlan(), 2.0)[0]   # leaves over x = 40 in the 15th step


def _single_engine_failure_step():
    single = Simulation(LocationHash2D(40.0, 40.0, 2.0, (0.0, 0.0)))
    _runaway_scene(single)
    for k in range(40):
        try:
            single.step(0.05, report=True)
        except Exception as err:  # noqa: BLE001
            assert "Index out of bounds" in str(err)
            return k
    raise AssertionError("the runaway never left")


@pytest.mark.gpu
@pytest.mark.parametrize("with_report", [True, False])
def test_a_tile_whose_agent_leaves_the_grid_stops_the_whole_mesh(with_report):
    """lib.rs:299-302: step returns Err("Index out of bounds").  The single engine commits nothing and can go on; a
    mesh whose tile failed is stopped as a whole (the other tiles have taken the step): every later call reports the
    same error, nothing is exchanged between tiles that stand at different time steps."""
    grid = dict(width=40.0, height=40.0, cell_size=2.0, offset=(0.0, 0.0))
    mesh = NativeTileMesh(LocationHash2D(**grid), (2, 2), 1)
    _runaway_scene(mesh)
    failed_at = _single_engine_failure_step()
    assert failed_at == 14
    raised = None
    for k in range(60):
        try:
            mesh.step(0.05, report=with_report)
            if not with_report and k % 8 == 7:
                mesh.synchronize()
        except Exception as err:  # noqa: BLE001
            raised = (k, str(err))
            break
    assert raised is not None and "Index out of bounds" in raised[1]
    assert raised[0] == failed_at if with_report else failed_at <= raised[0] <= failed_at + 8
    for call in (lambda: mesh.step(0.05), lambda: mesh.read_agents(), lambda: mesh.add_agents([(5.0, 5.0)], StubHighLevelPlan((0, 0)), NoLocalPlan(), 1.0),
                 lambda: mesh.synchronize(), lambda: mesh.recut()):
        with pytest.raises(Exception, match="Index out of bounds"):
            call()


@pytest.mark.gpu
def test_a_failing_step_of_the_distributed_form_is_agreed_on(monkeypatch):
    """The distributed form over RCCL (here a communicator of one): steps made without a report do not wait for the
    device, so the failure is acted on at the agreed check (every CS_MESH_CHECK_EVERY steps) or in synchronize, never by
    leaving the schedule in the middle: until then the rank issues its steps' collectives and nothing else."""
    monkeypatch.setenv("CS_MESH_CHECK_EVERY", "8")
    grid = dict(width=40.0, height=40.0, cell_size=2.0, offset=(0.0, 0.0))
    uid = Simulation(LocationHash2D(**grid)).rccl_unique_id()
    mesh = NativeTileMesh(LocationHash2D(**grid), (1, 1), 1, rccl_unique_id=uid, rank=0, n_ranks=1)
    _runaway_scene(mesh)
    raised = None
    for k in range(60):
        try:
            mesh.step(0.05, report=False)
        except Exception as err:  # noqa: BLE001
            raised = (k, str(err))
            break
    assert raised is not None and "Index out of bounds" in raised[1]
    # at an agreed check, the first or second after the event (the single engine fails in step 14)
    assert raised[0] in (15, 23) and raised[0] >= _single_engine_failure_step()
    with pytest.raises(Exception, match="Index out of bounds"):
        mesh.step(0.05, report=True)


def _rank_failing(rank, world, port, out_path, layout=(2, 1), second=False):
    import os
    import pickle
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from rmf_crowdsim_amd.tiles import TorchHostTransport
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = dict(width=40.0, height=40.0, cell_size=2.0, offset=(0.0, 0.0))
        mesh = NativeTileMesh(LocationHash2D(**grid), layout, 1, device=0, rank=rank, n_ranks=world,
                              host_transport=TorchHostTransport(dist))
        _runaway_scene(mesh)   # the runaway lives on the last rank's tile (x >= 20, or >= 30)
        if second:  # a second runaway, on the FIRST rank's tile, leaves over the top (y = 40) two steps after the first
            mesh.add_agents([(5.0, 36.6)], StubHighLevelPlan((0.0, 4.0)), NoLocalPlan(), 2.0)
        raised = None
        for k in range(60):
            try:
                mesh.step(0.05, report=False)
            except Exception as err:  # noqa: BLE001
                raised = (k, str(err))
                break
        again = None
        try:
            mesh.step(0.05, report=False)
        except Exception as err:  # noqa: BLE001
            again = str(err)
        with open(f"{out_path}.{rank}", "wb") as f:
            pickle.dump((raised, again), f)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_agree_on_a_failure_and_nobody_hangs(tmp_path):
    """One rank's tile fails; both ranks leave the schedule at the SAME step, the failing one with the reference's
    error, the other with "a tile of this mesh failed ... on another rank"; nobody waits in an exchange for a peer
    that has gone."""
    import pickle
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "failing")
    procs = [ctx.Process(target=_rank_failing, args=(r, 2, 29751, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, "a rank was left waiting for a peer that had gone"
    assert all(p.exitcode == 0 for p in procs)
    got = [pickle.load(open(f"{out}.{r}", "rb")) for r in range(2)]
    (r0, again0), (r1, again1) = got
    assert r0 is not None and r1 is not None and r0[0] == r1[0] and 14 <= r0[0] <= 22
    assert "Index out of bounds" in r1[1] and "another rank" in r0[1]
    assert again0 == r0[1] and again1 == r1[1]


def _failing_ranks(tmp_path, world, layout, port, second=False):
    import pickle
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "failing")
    procs = [ctx.Process(target=_rank_failing, args=(r, world, port, out, layout, second)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, "a rank was left waiting for a peer that had gone"
    assert all(p.exitcode == 0 for p in procs)
    return [pickle.load(open(f"{out}.{r}", "rb")) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("world,layout", [(2, (2, 1)), (4, (4, 1))])
def test_ranks_agree_on_a_failure_through_the_halo_headers_alone(tmp_path, monkeypatch, world, layout):
    """Review of round 4: without a report the ranks of a distributed mesh learnt of a failure at an all-reduce every 32
    steps.  Now the failure word travels in the halo headers every neighbour receives every step anyway (no collective):
    with the backstop all-reduce switched off (CS_MESH_CHECK_EVERY = 1,000,000) every rank leaves at the SAME call, within
    tiles_x + tiles_y steps of the failing one (the single engine fails in step 14), the failing rank with the
    reference's error, the others with "another rank", and nobody is left waiting.  4 x 1: the word crosses three tiles."""
    monkeypatch.setenv("CS_MESH_CHECK_EVERY", "1000000")
    got = _failing_ranks(tmp_path, world, layout, 29753 + world)
    steps = [g[0][0] for g in got]
    assert all(g[0] is not None for g in got) and len(set(steps)) == 1
    assert 14 < steps[0] <= 14 + layout[0] + layout[1]
    assert "Index out of bounds" in got[-1][0][1] and all("another rank" in g[0][1] for g in got[:-1])
    assert all(g[1] == g[0][1] for g in got)


@pytest.mark.gpu
def test_two_failures_on_two_ranks_within_the_agreement_window(tmp_path, monkeypatch):
    """4 x 1 tiles, one rank each: the last tile fails in step 14, the first in step 17 (its runaway leaves over the top),
    i.e. while the first failure's word is still on its way across the mesh.  The smallest word wins on every tile, so all
    four ranks leave at the call the FIRST failure fixes, nobody issues an exchange the others will not take, both failing
    ranks report the reference's error and the two in between "another rank"."""
    monkeypatch.setenv("CS_MESH_CHECK_EVERY", "1000000")
    got = _failing_ranks(tmp_path, 4, (4, 1), 29767, second=True)
    steps = [g[0][0] for g in got]
    assert all(g[0] is not None for g in got) and len(set(steps)) == 1 and 14 < steps[0] <= 14 + 5
    assert "Index out of bounds" in got[3][0][1] and "Index out of bounds" in got[0][0][1]
    assert "another rank" in got[1][0][1] and "another rank" in got[2][0][1]


@pytest.mark.gpu
def test_the_backstop_alone_still_agrees(tmp_path, monkeypatch):
    """CS_MESH_HEADER_AGREE=0: only the all-reduce every CS_MESH_CHECK_EVERY steps (round 4's agreement)."""
    monkeypatch.setenv("CS_MESH_HEADER_AGREE", "0")
    monkeypatch.setenv("CS_MESH_CHECK_EVERY", "8")
    got = _failing_ranks(tmp_path, 2, (2, 1), 29759)
    assert got[0][0][0] == got[1][0][0] and 14 <= got[0][0][0] <= 14 + 9
    assert "Index out of bounds" in got[1][0][1] and "another rank" in got[0][0][1]


@pytest.mark.gpu
@pytest.mark.parametrize("tiling", [(2, 2), (3, 1)])
def test_an_agent_the_index_refused_on_a_mesh(tiling):
    """lib.rs:133-149 on a mesh: the agents before the refused one stay added ON EVERY TILE (the loop over the tiles used
    to leave at the first tile that reported the failure: the tiles behind it missed them and their id counters fell
    behind), the refused agent is counted, read and removable, every step fails until then without poisoning the mesh:
    the single engine's story (== the oracle's, tests/test_gpu_parity.py) call for call."""
    from test_gpu_parity import _refused_agent_story
    grid = dict(width=100.0, height=100.0, cell_size=2.0, offset=(0.0, 0.0))
    mesh = NativeTileMesh(LocationHash2D(**grid), tiling, 1)
    single = Simulation(LocationHash2D(**grid))
    got, want = _refused_agent_story(mesh), _refused_agent_story(single)
    assert got == want and want[3][0] == [0, 2]
    # ... and the crowd goes on, on both, alike: the agent in tile (0, 0) and the one added after the failure
    for t in (mesh, single):
        t.add_agents([(49.9, 49.9), (52.0, 50.5)], StubHighLevelPlan((0.5, 0.25)), Zanlungo(*scenes.METRIC_ZANLUNGO), 3.0)
        for _ in range(30):
            t.step(0.05)
    assert mesh.read_agents().tobytes() == single.read_agents().tobytes() and len(mesh) == 4
