"""The window builder off the critical path (small crowds; csrc/cs_engine.hip.inc `windows_shadow`): the builder is a
~10 us chain of dependent searches that a 125k-agent step used to wait for between its scatter and its neighbour kernel.
A small crowd now steps on the windows of the step BEFORE, while this step's windows are cut by workgroups of their own
inside the neighbour kernel's launch; the windows tile their band completely (filler windows over the columns nobody
owns yet), so agents that walk, are spawned or are added anywhere still find a workgroup.  Which windows step an agent
changes nothing it computes: windows one step old, windows built in every step's own launch (CS_WINDOWS_KEEP=0) and the
exact gather kernel must give the same bits."""
import numpy as np
import pytest

from rmf_crowdsim_amd import (LocationHash2D, MonotonicCrowd, NoLocalPlan, Simulation, SourceSink, StubHighLevelPlan,
                              Zanlungo, _abi, scenes)
from rmf_crowdsim_amd.tiles import NativeTileMesh

pytestmark = pytest.mark.gpu
LP = Zanlungo(*scenes.METRIC_ZANLUNGO)


def _three_ways(monkeypatch, build, steps, report_every=25):
    runs = {}
    for name, keep, flags in (("kept", "1", 2), ("every", "0", 2), ("gather", "0", 1)):
        monkeypatch.setenv("CS_WINDOWS_KEEP", keep)
        sim = build(flags)
        for k in range(steps):
            sim.step(0.05, report=(k % report_every == report_every - 1))
            hook = getattr(sim, "_between_steps", None)
            if hook:
                hook(k)
        runs[name] = (sim.read_agents(), sim.kernel_stat(_abi.CS_STAT_STEPS_ON_KEPT_WINDOWS), dict(sim.last_report),
                      sim.kernel_stat(_abi.CS_STAT_WINDOWS_OFF_LDS))
    assert runs["kept"][1] >= (steps * 2) // 3 and runs["every"][1] == 0 and runs["gather"][1] == 0
    a = runs["kept"][0]
    assert a.tobytes() == runs["every"][0].tobytes() == runs["gather"][0].tobytes()
    assert runs["kept"][2] == runs["every"][2]
    return runs


@pytest.mark.parametrize("n,axis", [(125000, 0), (40000, 1), (6000, 1)])
def test_walkers_entering_empty_rows_and_columns(n, axis, monkeypatch):
    """The crowd walks at 1.3 m/s along x (rows: it enters empty BANDS ahead of it) or along y (columns: it enters the
    filler windows at the end of every band) for 150 steps = 9.75 m, i.e. five cells beyond where the windows of the
    first step ended."""
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=3, cell_size=2.0, room=20.0)
    if axis == 1:  # the same crowd, walking +y: room on the high-y side instead
        grid = dict(grid, width=grid["height"], height=grid["width"], offset=(grid["offset"][1], grid["offset"][0]))
        pts = pts[:, ::-1].copy()

    def build(flags):
        sim = Simulation(LocationHash2D(**grid), flags=flags)
        walk = (scenes.WALK_SPEED, 0.0) if axis == 0 else (0.0, scenes.WALK_SPEED)
        creep = scenes.CREEP_SPEED
        for g, sgn in ((0, 1.0), (1, -1.0)):
            v = (walk[0] + (0.0 if axis == 0 else sgn * creep), walk[1] + (sgn * creep if axis == 0 else 0.0))
            sim.add_agents(pts[group == g], StubHighLevelPlan(v), LP, 2.0)
        return sim
    runs = _three_ways(monkeypatch, build, 150)
    a = runs["kept"][0]
    assert len(a) == n and np.isfinite(a["x"]).all() and runs["kept"][2]["n_nonfinite"] == 0
    print(f"kept windows, {n} agents walking along axis {axis}: windows off the LDS path {runs['kept'][3]} (rebuilt every step: {runs['every'][3]})")
    assert runs["kept"][3] <= 40   # (a handful of windows at the crowd's moving front outgrow their LDS tile for a step: slower, not wrong)


def test_agents_added_and_spawned_where_no_window_owned_anything(monkeypatch):
    """A block of agents dropped into an empty corner of the grid between two steps on kept windows, and source-sinks
    whose sources lie in empty cells: the filler windows step them from their first step on."""
    pts, grid, extent, group = scenes.uniform_crowd(30000, seed=8, cell_size=2.0, room=0.0)
    grid = dict(grid, width=grid["width"] + 60.0, height=grid["height"] + 60.0)
    late = scenes.jittered_lattice(900, 0.63, (grid["offset"][0] + grid["height"] - 40.0, grid["offset"][1] + grid["width"] - 40.0), 0.2, 6)

    def build(flags):
        sim = Simulation(LocationHash2D(**grid), flags=flags)
        scenes.add_counterflow(sim, pts, group, scenes.CREEP_SPEED, LP, 2.0)
        x0 = grid["offset"][0] + grid["height"] - 20.0
        for k in range(6):
            y = grid["offset"][1] + 6.0 + 3.0 * k
            sim.add_source_sink(SourceSink((x0, y), 0.5, MonotonicCrowd(1000.0), StubHighLevelPlan((-1.3, 0.0)), LP,
                                           [(x0 - 12.0, y)], False, 2.0))

        def between(k):
            if k == 5:   # on a step that runs on kept windows (built at steps 0 and 4)
                sim.add_agents(late, StubHighLevelPlan((0.01, 0.0)), LP, 2.0)
        sim._between_steps = between
        return sim
    runs = _three_ways(monkeypatch, build, 60, report_every=7)
    a = runs["kept"][0]
    assert len(a) > 30000 + 900 and runs["kept"][2]["n_spawned"] >= 0


def test_kept_windows_on_the_tiles_of_a_mesh(monkeypatch):
    """A 2 x 2 mesh of ~30k-agent tiles walking over the cuts: every tile keeps its windows (ghost rows, owned
    rectangles, a tile's owned count from the scatter); the mesh equals one engine bit for bit, and says how many agents
    it holds at every report."""
    n, steps = 120000, 90
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=9, cell_size=2.0, room=scenes.WALK_SPEED * 0.05 * (steps + 8) + 4.0)
    monkeypatch.setenv("CS_WINDOWS_KEEP", "0")
    single = Simulation(LocationHash2D(**grid), flags=1)
    monkeypatch.setenv("CS_WINDOWS_KEEP", "1")
    mesh = NativeTileMesh(LocationHash2D(**grid), (2, 2), 1, density_per_cell=15.0, weights=pts)
    for t in (single, mesh):
        scenes.add_walking_crowd(t, pts, group, LP, 2.0)
    for k in range(steps):
        with_report = k % 10 == 9
        single.step(0.05, report=with_report)
        mesh.step(0.05, report=with_report)
        if with_report:
            assert mesh.last_report["n_agents"] == single.last_report["n_agents"] == n
    assert mesh.tile(0).kernel_stat(_abi.CS_STAT_STEPS_ON_KEPT_WINDOWS) >= steps // 2
    assert single.read_agents().tobytes() == mesh.read_agents().tobytes()
    assert int(mesh.tile_counts().sum()) == n


def test_queries_reads_and_removals_between_steps_leave_the_kept_windows_alone(monkeypatch):
    """A radius query, a read-back or a removal between two steps sorts the agents outside any step: that sort's scan
    used to clear the window count of the very list the next step was about to run on (nobody stepped: the crowd was
    gone a step later), and on a tile it zeroed the owned count with no builder behind it to count again."""
    pts, grid, extent, group = scenes.uniform_crowd(20000, seed=12, cell_size=2.0, room=4.0)

    def build(flags):
        sim = Simulation(LocationHash2D(**grid), flags=flags)
        scenes.add_counterflow(sim, pts, group, scenes.CREEP_SPEED, LP, 2.0)
        mid = (float(pts[:, 0].mean()), float(pts[:, 1].mean()))

        def between(k):
            if k % 7 == 3:
                assert len(sim.get_neighbours_in_radius(3.0, mid)) > 5
            if k % 7 == 5:
                assert len(sim.read_agents()) == len(sim)
            if k == 20:
                sim.remove_agents(17)
                sim.remove_agents(4242)
        sim._between_steps = between
        return sim
    runs = _three_ways(monkeypatch, build, 45, report_every=15)
    assert len(runs["kept"][0]) == len(pts) - 2


def test_a_tile_nobody_stands_in_steps_on_fillers_alone(monkeypatch):
    """The crowd stands in one tile of a 2 x 2 mesh; the slot count of the three empty tiles is a host-side upper bound
    (what the halo buffers could deliver), large enough for the tiled kernel: their kept lists are fillers only, over
    sorted arrays that hold no record.  (Those windows used to load 'slot 0' for their idle lanes and index the group
    table with whatever lay there: a memory fault on an unlucky day.)"""
    from rmf_crowdsim_amd.tiles import LocalTileMesh
    n = 3200
    pts = scenes.jittered_lattice(n, 0.75, (6.0, 96.0), 0.2, 5)
    grid = dict(width=174.0, height=174.0, cell_size=3.0, offset=(0.0, 0.0))
    monkeypatch.setenv("CS_CHECK_WINDOWS", "1")
    single = Simulation(LocationHash2D(**grid))
    mesh = LocalTileMesh(LocationHash2D(**grid), (2, 2), halo_cells=1, phases=1)
    for t in (single, mesh):
        t.add_agents(pts[: n // 2], StubHighLevelPlan((0.2, 0.3)), LP, 1.0)
        t.add_agents(pts[n // 2:], StubHighLevelPlan((-0.3, 0.1)), LP, 0.8)
    for _ in range(12):
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    counts = [len(e) for e in mesh.engines]
    assert sorted(counts)[:3] == [0, 0, 0] and sum(counts) == n
    assert sum(e.kernel_stat(_abi.CS_STAT_STEPS_ON_KEPT_WINDOWS) for e in mesh.engines) >= 20  # (the empty tiles too)
    assert single.read_agents().tobytes() == mesh.read_agents().tobytes()


def test_one_pass_scan_and_two_launch_scan_sort_alike(monkeypatch):
    """The cell scan is ONE launch (`k_scan_onepass`: a tile adds up the totals of the tiles before it as they are
    published) for grids of up to 1,024 scan tiles and the two-launch form beyond; `CS_SCAN_ONEPASS=0` selects the
    latter everywhere.  Same bytes either way, on a grid of some sixty scan tiles with a walking crowd (cells change hands every
    step) and source-sinks (slots beyond the live records)."""
    pts, grid, extent, group = scenes.uniform_crowd(150000, seed=4, cell_size=1.0, room=6.0)
    outs = []
    for onepass in ("1", "0"):
        monkeypatch.setenv("CS_SCAN_ONEPASS", onepass)
        sim = Simulation(LocationHash2D(**grid))
        scenes.add_walking_crowd(sim, pts, group, LP, 2.0)
        x0, y0 = grid["offset"][0] + 3.0, grid["offset"][1] + 3.0
        for k in range(4):
            sim.add_source_sink(SourceSink((x0, y0 + 2.0 * k), 0.5, MonotonicCrowd(1000.0), StubHighLevelPlan((0.0, 1.0)), LP,
                                           [(x0, y0 + 2.0 * k + 4.0)], False, 2.0))
        for k in range(40):
            sim.step(0.05, report=(k % 13 == 12))
        outs.append(sim.read_agents())
    assert len(outs[0]) > 150000 and outs[0].tobytes() == outs[1].tobytes()
