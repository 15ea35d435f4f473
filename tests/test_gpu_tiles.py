"""Multi-GPU logic on one GPU: the same crowd stepped by one engine and by a mesh of tile
engines (ghost rings + two-phase halo exchange) must agree bit for bit."""
import numpy as np
import pytest

from rmf_crowdsim_amd import LocationHash2D, NoLocalPlan, Simulation, StubHighLevelPlan, Zanlungo, scenes
from rmf_crowdsim_amd.tiles import LocalTileMesh

pytestmark = pytest.mark.gpu


def _populate(target, pts, group, velocities, lp, eyesight):
    for g, v in enumerate(velocities):
        target.add_agents(pts[group == g], StubHighLevelPlan(v), lp, eyesight)


@pytest.mark.parametrize("tiles", [(2, 2), (3, 1), (1, 2)])
def test_migration_across_tiles_matches_single_engine(tiles):
    # diagonal walkers, no local planner: 300 steps * 0.065 m = ~20 m = 10 cells of travel,
    # so a large share of the crowd changes tile (some through a corner)
    n = 6000
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=21, cell_size=2.0, margin=30.0)
    vel = [(0.9, 0.9), (-0.9, 0.6)]
    single = Simulation(LocationHash2D(**grid))
    mesh = LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=1)
    for t in (single, mesh):
        _populate(t, pts, group, vel, NoLocalPlan(), 2.0)
    for _ in range(300):
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    a, b = single.read_agents(), mesh.read_agents()
    assert len(b) == n and len(mesh) == n
    assert a.tobytes() == b.tobytes()
    owners = [len(e) for e in mesh.engines]
    print("tile populations", owners)
    assert min(owners) > 0


@pytest.mark.parametrize("tiles,cell,eyesight,halo", [((2, 2), 2.0, 2.0, 1), ((2, 2), 1.0, 2.0, 2),
                                                       ((4, 2), 2.0, 2.0, 1)])
def test_zanlungo_across_tiles_matches_single_engine(tiles, cell, eyesight, halo):
    n = 30000
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=5, cell_size=cell, margin=20.0)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    # co-flow with a 2 cm/s speed difference: everybody crosses cells and tiles, nobody can
    # close a lattice gap within the run, every agent has neighbours with finite t_i
    vel = [(1.30, 0.4), (1.28, 0.4)]
    single = Simulation(LocationHash2D(**grid))
    mesh = LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=halo)
    for t in (single, mesh):
        _populate(t, pts, group, vel, lp, eyesight)
    for k in range(60):
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    single.step(0.05)
    mesh.step(0.05)
    a, b = single.read_agents(), mesh.read_agents()
    assert single.last_report["n_tti_zero"] == 0
    assert len(b) == n
    assert a.tobytes() == b.tobytes()


def test_creeping_counterflow_across_tiles():
    n = 30000
    pts, grid, extent, group = scenes.uniform_crowd(n, seed=9, cell_size=2.0, margin=10.0)
    lp = Zanlungo(*scenes.METRIC_ZANLUNGO)
    single = Simulation(LocationHash2D(**grid))
    mesh = LocalTileMesh(LocationHash2D(**grid), (2, 2), halo_cells=1)
    for t in (single, mesh):
        scenes.add_counterflow(t, pts, group, scenes.CREEP_SPEED, lp, 2.0)
    for k in range(50):
        single.step(0.05, report=False)
        mesh.step(0.05, report=False)
    a, b = single.read_agents(), mesh.read_agents()
    force = np.hypot(a["vx"], np.abs(a["vy"]) - scenes.CREEP_SPEED)
    assert np.mean(force > 0) > 0.9
    assert a.tobytes() == b.tobytes()
