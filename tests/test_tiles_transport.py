"""CPU tests of the tile layout and of the two-phase halo transport (world_size 2, gloo)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rmf_crowdsim_amd import LocationHash2D
from rmf_crowdsim_amd.tiles import (OPPOSITE, RECORD, XHI, XLO, YHI, YLO, TileLayout, default_tiling,
                                    ALL_DIRS, exchange_all, exchange_axis, halo_capacity)


def test_layout_covers_the_grid_without_overlap():
    idx = LocationHash2D(400.0, 200.0, 2.0, (0.0, 0.0))
    lay = TileLayout(idx, 4, 2)
    assert (lay.rows, lay.cols) == (100, 200)
    seen = np.zeros((lay.rows, lay.cols), dtype=int)
    for t in range(lay.n_tiles):
        x0, x1, y0, y1 = lay.rect(*lay.coords(t))
        seen[x0:x1, y0:y1] += 1
    assert (seen == 1).all()
    assert lay.neighbour(0, 0, XLO) is None and lay.neighbour(0, 0, XHI) == lay.index(1, 0)
    assert lay.neighbour(3, 1, YLO) == lay.index(3, 0) and lay.neighbour(3, 1, YHI) is None
    for t in range(lay.n_tiles):
        for d in ALL_DIRS:  # edges and corners
            peer = lay.neighbour(*lay.coords(t), d)
            if peer is not None:
                assert lay.neighbour(*lay.coords(peer), OPPOSITE[d]) == t
    assert default_tiling(8) == (4, 2) and default_tiling(4) == (2, 2) and default_tiling(2) == (2, 1)
    assert halo_capacity(lay, 10.0, 1) >= 1024


def _worker_all(rank, world, port, tiles, results):
    """One-phase exchange: every rank posts all its sends and receives (edges and corners) at once."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = TileLayout(LocationHash2D(64.0, 64.0, 2.0, (0.0, 0.0)), *tiles)
        tx, ty = lay.coords(rank)
        cap = 8
        bufs = {}
        for d in ALL_DIRS:
            if lay.neighbour(tx, ty, d) is None:
                continue
            send = torch.full(((cap + 1) * RECORD,), 16 * rank + d, dtype=torch.uint8)
            recv = torch.zeros((cap + 1) * RECORD, dtype=torch.uint8)
            bufs[d] = (send, recv)
        for _ in range(2):  # twice: buffers are reused every step
            exchange_all(dist, lay, rank, bufs)
        ok = True
        for d, (send, recv) in bufs.items():
            peer = lay.neighbour(tx, ty, d)
            ok = ok and bool((recv == 16 * peer + OPPOSITE[d]).all())
        results[rank] = (ok, len(bufs))
    finally:
        dist.destroy_process_group()


def test_four_rank_one_phase_exchange_over_gloo():
    """2 x 2 tiles: every rank has two edge neighbours and one diagonal neighbour."""
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_all, args=(r, 4, 29671, (2, 2), results)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(results) == {r: (True, 3) for r in range(4)}


def _worker(rank, world, port, tiles, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = TileLayout(LocationHash2D(64.0, 64.0, 2.0, (0.0, 0.0)), *tiles)
        tx, ty = lay.coords(rank)
        cap = 8
        bufs = {}
        for d in (XLO, XHI, YLO, YHI):
            if lay.neighbour(tx, ty, d) is None:
                continue
            send = torch.full(((cap + 1) * RECORD,), 16 * rank + d, dtype=torch.uint8)
            recv = torch.zeros((cap + 1) * RECORD, dtype=torch.uint8)
            bufs[d] = (send, recv)
        for axis in (0, 1):
            exchange_axis(dist, lay, rank, bufs, axis)
        ok = True
        for d, (send, recv) in bufs.items():
            peer = lay.neighbour(tx, ty, d)
            ok = ok and bool((recv == 16 * peer + OPPOSITE[d]).all())
        results[rank] = ok and len(bufs) > 0
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tiles", [(2, 1), (1, 2)])
def test_two_rank_exchange_over_gloo(tiles):
    ctx = mp.get_context("spawn")
    results = ctx.Manager().dict()
    port = 29650 + tiles[0]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tiles, results)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(results) == {0: True, 1: True}


def test_weighted_cuts_balance_a_clustered_crowd():
    """BASELINE.json configs[4]: cuts at the quantiles of the row / column histograms."""
    from rmf_crowdsim_amd import scenes
    pts, grid, extent, group = scenes.hotspot_crowd(120_000, seed=3)
    si = LocationHash2D(**grid)
    even = TileLayout(si, 4, 2)
    weighted = TileLayout(si, 4, 2, weights=pts, min_cells=4)
    ce, cw = even.tile_counts(pts, si), weighted.tile_counts(pts, si)
    assert ce.sum() == cw.sum() == len(pts)
    assert cw.max() / cw.mean() < ce.max() / ce.mean() and cw.max() / cw.mean() < 1.2
    assert weighted.x_edges[0] == 0 and weighted.x_edges[-1] == weighted.rows
    assert min(np.diff(weighted.x_edges)) >= 4 and min(np.diff(weighted.y_edges)) >= 4
    # nobody overlaps: the lattice spacing exceeds the agent radius
    cell = np.floor(pts / 0.2).astype(np.int64)
    assert len(np.unique(cell[:, 0] * 100000 + cell[:, 1])) == len(pts)
