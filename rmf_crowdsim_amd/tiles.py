"""Spatial tiles: one engine per GPU, a ghost ring per tile, a two-phase halo exchange.

The reference is a single process; this is the multi-GPU row of the hot path.
The global LocationHash2D grid is cut into tiles_x x tiles_y rectangles of whole cells.  Every
tile engine owns the agents whose cell lies in its rectangle and keeps `halo_cells` ghost
cells around it.  Per step:

    pack -> one exchange with the (up to) eight neighbours -> unpack       (cs_halo_pack_all /
    cs_step on every tile                                                   cs_halo_unpack_all)

(phases=2 selects the older two-phase schedule: pack X -> exchange -> unpack X -> pack Y ->
exchange -> unpack Y, where Y forwards what arrived in X so that corners reach diagonal tiles;
same result, twice the transport rounds.)

Two transports move the fixed-capacity device buffers:
    LocalTileMesh      several tiles in ONE process on one GPU (device-to-device copies);
                       the test double of the multi-GPU logic on a single-GPU box
    DistributedTiles   one rank per GPU under torch.distributed (backend "nccl" = RCCL over
                       xGMI; "gloo" with CPU tensors in the transport tests)
"""
import numpy as np

from . import _abi
from .simulation import LocationHash2D, Simulation

XLO, XHI, YLO, YHI = _abi.CS_DIR_XLO, _abi.CS_DIR_XHI, _abi.CS_DIR_YLO, _abi.CS_DIR_YHI
XLO_YLO, XLO_YHI, XHI_YLO, XHI_YHI = (_abi.CS_DIR_XLO_YLO, _abi.CS_DIR_XLO_YHI, _abi.CS_DIR_XHI_YLO,
                                      _abi.CS_DIR_XHI_YHI)
EDGES = (XLO, XHI, YLO, YHI)
ALL_DIRS = EDGES + (XLO_YLO, XLO_YHI, XHI_YLO, XHI_YHI)
OPPOSITE = {XLO: XHI, XHI: XLO, YLO: YHI, YHI: YLO, XLO_YLO: XHI_YHI, XHI_YHI: XLO_YLO,
            XLO_YHI: XHI_YLO, XHI_YLO: XLO_YHI}
STEP_OF = {XLO: (-1, 0), XHI: (1, 0), YLO: (0, -1), YHI: (0, 1), XLO_YLO: (-1, -1), XLO_YHI: (-1, 1),
           XHI_YLO: (1, -1), XHI_YHI: (1, 1)}
RECORD = _abi.CS_HALO_RECORD_BYTES


def _weighted_edges(hist, parts, min_width):
    """Cut range(len(hist)) into `parts` runs of about equal weight, each at least `min_width`
    cells wide (the cumulative histogram's quantiles, pushed apart where needed)."""
    n = len(hist)
    cum = np.concatenate([[0.0], np.cumsum(hist, dtype=np.float64)])
    total = cum[-1]
    edges = [0]
    for k in range(1, parts):
        e = int(np.searchsorted(cum, total * k / parts, side="left"))
        e = max(e, edges[-1] + min_width)
        e = min(e, n - (parts - k) * min_width)
        edges.append(e)
    edges.append(n)
    return edges


class TileLayout:
    """Tensor-product split of the global cell grid: tiles_x runs of rows by tiles_y runs of
    columns.  `x` is the index that location_to_index multiplies by the row stride
    (location_hash_2d.rs:59); there are (height / cell) x-rows of (width / cell) cells each.
    Without `weights` the runs are even.  With `weights` (agent positions, n x 2) the cuts are
    the quantiles of the per-row and per-column agent histograms, so a clustered crowd
    (BASELINE.json configs[4]) is spread more evenly; runs stay at least `min_cells` wide."""

    def __init__(self, spatial_index, tiles_x, tiles_y, weights=None, min_cells=2, histograms=None):
        self.cols = int(spatial_index.width / spatial_index.cell_size)   # stride, y cells per row
        self.rows = int(spatial_index.height / spatial_index.cell_size)  # x rows
        self.tiles_x, self.tiles_y = int(tiles_x), int(tiles_y)
        if histograms is not None:  # (agents per x-row, agents per y-column): cs_tile_histogram over a mesh
            self.x_edges = _weighted_edges(np.asarray(histograms[0], dtype=np.float64), self.tiles_x, int(min_cells))
            self.y_edges = _weighted_edges(np.asarray(histograms[1], dtype=np.float64), self.tiles_y, int(min_cells))
        elif weights is None:
            self.x_edges = [round(k * self.rows / self.tiles_x) for k in range(self.tiles_x + 1)]
            self.y_edges = [round(k * self.cols / self.tiles_y) for k in range(self.tiles_y + 1)]
        else:
            w = np.asarray(weights, dtype=np.float64)
            off = spatial_index.offset
            cx = np.clip(np.floor((w[:, 0] - off[0]) / spatial_index.cell_size), 0, self.rows - 1)
            cy = np.clip(np.floor((w[:, 1] - off[1]) / spatial_index.cell_size), 0, self.cols - 1)
            self.x_edges = _weighted_edges(np.bincount(cx.astype(np.int64), minlength=self.rows),
                                           self.tiles_x, int(min_cells))
            self.y_edges = _weighted_edges(np.bincount(cy.astype(np.int64), minlength=self.cols),
                                           self.tiles_y, int(min_cells))

    def tile_counts(self, positions, spatial_index):
        """Agents per tile (tiles_x x tiles_y) for a set of positions: the imbalance report."""
        w = np.asarray(positions, dtype=np.float64)
        off = spatial_index.offset
        cx = np.clip(np.floor((w[:, 0] - off[0]) / spatial_index.cell_size), 0, self.rows - 1)
        cy = np.clip(np.floor((w[:, 1] - off[1]) / spatial_index.cell_size), 0, self.cols - 1)
        tx = np.searchsorted(np.asarray(self.x_edges[1:-1]), cx, side="right")
        ty = np.searchsorted(np.asarray(self.y_edges[1:-1]), cy, side="right")
        return np.bincount(tx * self.tiles_y + ty, minlength=self.n_tiles).reshape(self.tiles_x, self.tiles_y)

    @property
    def n_tiles(self):
        return self.tiles_x * self.tiles_y

    def coords(self, index):
        return index // self.tiles_y, index % self.tiles_y

    def index(self, tx, ty):
        return tx * self.tiles_y + ty

    def rect(self, tx, ty):
        return (self.x_edges[tx], self.x_edges[tx + 1], self.y_edges[ty], self.y_edges[ty + 1])

    def neighbour(self, tx, ty, direction):
        dx, dy = STEP_OF[direction]
        nx, ny = tx + dx, ty + dy
        if 0 <= nx < self.tiles_x and 0 <= ny < self.tiles_y:
            return self.index(nx, ny)
        return None

    def min_tile_cells(self):
        return min(min(np.diff(self.x_edges)), min(np.diff(self.y_edges)))


def default_tiling(n):
    """4 x 2 for 8 GPUs (BASELINE.json configs[2]), the squarest split otherwise."""
    best = (n, 1)
    for a in range(1, n + 1):
        if n % a == 0 and a >= n // a:
            best = (a, n // a)
            break
    return best


def halo_capacity(layout, density_per_cell, halo_cells, slack=2.0):
    """Records per direction, the same on every tile (both ends of a link must agree): a band
    of 2*halo cells along the longest tile edge, with slack."""
    edge = max(max(np.diff(layout.x_edges)), max(np.diff(layout.y_edges))) + 4 * halo_cells
    return int(max(1024, slack * density_per_cell * edge * 2 * halo_cells))


def halo_capacity_of(layout, tx, ty, direction, density_per_cell, halo_cells, slack=2.0):
    """Records of ONE direction's buffers; both ends of a link compute the same number (the cuts are a
    tensor product: the tiles on either side of an edge have the same extent along it).  An edge
    carries a band of 2 * halo cells along the shared edge, a corner the (2 * halo)^2 cells where two
    bands cross: at 1M agents per tile the four corner buffers are a few KB instead of a megabyte
    each, and the whole exchange a third of what one capacity for all eight directions made it."""
    x0, x1, y0, y1 = layout.rect(tx, ty)
    band = 2 * halo_cells
    if direction in (XLO, XHI):
        cells = (y1 - y0 + 4 * halo_cells) * band
    elif direction in (YLO, YHI):
        cells = (x1 - x0 + 4 * halo_cells) * band
    else:
        cells = 4 * band * band
    return int(max(256, slack * density_per_cell * cells))


def _is_data_planner(source_sink):
    """True when the sink's high-level planner runs on the device without host events.  A route
    follower on a tile qualifies: its routes are planned when the sink is registered."""
    from .simulation import RouteFollower, _DataPlan
    return isinstance(source_sink.high_level_planner, (_DataPlan, RouteFollower))


def _has_route_legs(source_sink):
    """A route follower's sink with several waypoints: legs after the first start wherever the agent
    stands, so their (start, goal) pairs can miss the route book (cs_route_misses)."""
    from .simulation import RouteFollower
    return isinstance(source_sink.high_level_planner, RouteFollower) and len(source_sink.waypoints) > 1


def merge_radius_answers(parts):
    """Answers of the tiles of a mesh to one radius query -> the reference's list: cells x-major /
    y-minor (the global cell number is x * stride + y), ascending id inside a cell.  parts: (ids,
    squared distances, global cells) per tile, each tile reporting the agents it owns."""
    ids = np.concatenate([p[0] for p in parts])
    cells = np.concatenate([p[2] for p in parts])
    order = np.lexsort((ids, cells))
    return [int(v) for v in ids[order]]


def merge_knn_answers(parts, k):
    """k nearest over the tiles: every tile's k nearest owned agents, merged by (distance, id)."""
    ids = np.concatenate([p[0] for p in parts])
    d2 = np.concatenate([p[1] for p in parts])
    order = np.lexsort((ids, d2))[:k]
    return [int(v) for v in ids[order]]


class _TileBase:
    def _make_engine(self, spatial_index, layout, index, halo_cells, device, stream, capacity_hint,
                     flags):
        tx, ty = layout.coords(index)
        return Simulation(spatial_index, device=device, flags=flags, capacity_hint=capacity_hint,
                          stream=stream, tile=layout.rect(tx, ty), halo_cells=halo_cells)

    @staticmethod
    def _alloc(torch, n_records, device):
        return torch.zeros((n_records + 1) * RECORD, dtype=torch.uint8, device=device)


class LocalTileMesh(_TileBase):
    """All tiles of a layout in one process on one GPU.  Exchanges are device-to-device copies
    on the shared stream.  Same engine code path as one-rank-per-GPU."""

    def __init__(self, spatial_index, tiles, halo_cells, device=0, capacity_records=None,
                 density_per_cell=16.0, flags=0, weights=None, phases=1, capacity_hint=0):
        import torch
        self.torch = torch
        self.phases = int(phases)
        self.layout = TileLayout(spatial_index, *tiles, weights=weights, min_cells=2 * int(halo_cells))
        self.halo_cells = int(halo_cells)
        assert self.layout.min_tile_cells() >= 2 * self.halo_cells, "tiles thinner than two halos"
        dev = torch.device("cuda", device)
        self.bufs_device = dev
        # engines and buffer copies must share ONE real stream: torch's default stream has handle 0,
        # which the engine reads as "create your own", and then nothing orders pack -> copy -> unpack
        self.stream = torch.cuda.Stream(dev)
        stream = self.stream.cuda_stream
        self.spatial_index = spatial_index
        self._capacity_records, self._density_per_cell = capacity_records, density_per_cell
        self.engines, self.bufs = [], []
        for index in range(self.layout.n_tiles):
            sim = self._make_engine(spatial_index, self.layout, index, halo_cells, device, stream,
                                    capacity_hint, flags)
            self.engines.append(sim)
            self.bufs.append(self._set_buffers(sim, index))
        torch.cuda.synchronize(dev)  # the zero fills ran on the default stream

    def _set_buffers(self, sim, index):
        """Send / receive buffers of one tile towards its neighbours, sized for the current layout."""
        tx, ty = self.layout.coords(index)
        bufs = {}
        for d in (EDGES if self.phases == 2 else ALL_DIRS):
            if self.layout.neighbour(tx, ty, d) is None:
                continue
            cap = self._capacity_records or (
                halo_capacity(self.layout, self._density_per_cell, self.halo_cells) if self.phases == 2 else
                halo_capacity_of(self.layout, tx, ty, d, self._density_per_cell, self.halo_cells))
            send, recv = self._alloc(self.torch, cap, self.bufs_device), self._alloc(self.torch, cap, self.bufs_device)
            sim.halo_set_buffers(d, send.data_ptr(), recv.data_ptr(), cap)
            bufs[d] = (send, recv)
        return bufs

    def tile_counts(self):
        """Agents owned by each tile right now (tiles_x x tiles_y)."""
        return np.array([len(sim) for sim in self.engines]).reshape(self.layout.tiles_x, self.layout.tiles_y)

    def recut(self):
        """Move the cuts to the quantiles of where the crowd stands NOW (a clustered crowd drifts,
        BASELINE.json configs[4]): per-row / per-column agent histograms from the devices, a new
        tensor-product layout, every agent handed to the tile that owns its cell from now on.
        Between two steps; the state, and so every later step, is unchanged bit for bit."""
        rows = np.zeros(self.layout.rows, dtype=np.uint64)
        cols = np.zeros(self.layout.cols, dtype=np.uint64)
        for sim in self.engines:
            sim.tile_histogram(rows, cols)
        self.layout = TileLayout(self.spatial_index, self.layout.tiles_x, self.layout.tiles_y,
                                 min_cells=2 * self.halo_cells, histograms=(rows, cols))
        exports = [sim.tile_export() for sim in self.engines]
        self.torch.cuda.synchronize(self.bufs_device)
        for index, sim in enumerate(self.engines):
            sim.tile_retile(self.layout.rect(*self.layout.coords(index)))
            self.bufs[index] = self._set_buffers(sim, index)
        self.torch.cuda.synchronize(self.bufs_device)
        for sim in self.engines:
            for rec in exports:
                sim.tile_import(rec)
        return self.tile_counts()

    def add_agents(self, positions, high_level_planner, local_planner, eyesight):
        ids = None
        for sim in self.engines:  # every tile sees the global list and keeps what it owns
            ids = sim.add_agents(positions, high_level_planner, local_planner, eyesight)
        return ids

    def add_source_sink(self, source_sink):
        handle = None
        for sim in self.engines:  # every tile registers every sink; the owner of the source spawns
            handle = sim.add_source_sink(source_sink)
        self._has_sinks = True
        self._n_sinks = getattr(self, "_n_sinks", 0) + 1
        self._route_legs = getattr(self, "_route_legs", False) or _has_route_legs(source_sink)
        # (legs that may miss the route book need the host after the step: the host-side spawn path)
        self._host_planner = (getattr(self, "_host_planner", False) or not _is_data_planner(source_sink) or
                              self._route_legs)
        return handle

    def add_event_listener(self, listener):
        for sim in self.engines:
            sim.add_event_listener(listener)

    def remove_source_sink(self, handle):
        for sim in self.engines:  # same handles everywhere: every tile registered every sink
            sim.remove_source_sink(handle)

    def remove_agents(self, agent):
        """lib.rs:176-192.  Between steps every agent is held by exactly one tile (its owner)."""
        from .simulation import CrowdSimError
        for sim in self.engines:
            if sim.remove_agent_here(agent):  # an engine failure (not "not here") raises
                return
        raise CrowdSimError("unknown agent id")

    def _exchange(self, axis):
        for sim in self.engines:
            sim.halo_pack(axis)
        with self.torch.cuda.stream(self.stream):
            for index, bufs in enumerate(self.bufs):
                tx, ty = self.layout.coords(index)
                for d in ((XLO, XHI) if axis == 0 else (YLO, YHI)):
                    if d not in bufs:
                        continue
                    peer = self.layout.neighbour(tx, ty, d)
                    self.bufs[peer][OPPOSITE[d]][1].copy_(bufs[d][0], non_blocking=True)
        for sim in self.engines:
            sim.halo_unpack(axis)

    def _exchange_all(self):
        for sim in self.engines:
            sim.halo_pack_all()
        with self.torch.cuda.stream(self.stream):
            for index, bufs in enumerate(self.bufs):
                tx, ty = self.layout.coords(index)
                for d, (send, _) in bufs.items():
                    peer = self.layout.neighbour(tx, ty, d)
                    self.bufs[peer][OPPOSITE[d]][1].copy_(send, non_blocking=True)
        for sim in self.engines:
            sim.halo_unpack_all()

    def step(self, dur, report=True):
        if self.phases == 2:
            self._exchange(0)
            self._exchange(1)
        else:
            self._exchange_all()
        if getattr(self, "_has_sinks", False):
            if not report and not self._host_planner and not any(e.host_events_needed for e in self.engines):
                # nobody on the host listens: flags stay on the device (cs_spawn_probe_dev / _commit_dev)
                n = self.engines[0].source_sink_slots
                with self.torch.cuda.stream(self.stream):
                    if getattr(self, "_flag_bufs", None) is None or self._flag_bufs[0].numel() < n:
                        dev = self.bufs_device
                        self._flag_bufs = [self.torch.zeros(max(n, 1), dtype=self.torch.int32, device=dev)
                                           for _ in self.engines]
                    for sim, f in zip(self.engines, self._flag_bufs):
                        sim.spawn_probe_dev(dur, f.data_ptr(), n)
                    total = self._flag_bufs[0]
                    for f in self._flag_bufs[1:]:
                        total = self.torch.maximum(total, f)
                    for sim in self.engines:
                        sim.spawn_commit_dev(total.data_ptr(), n)
            else:
                flags = None
                for sim in self.engines:
                    f = sim.spawn_probe(dur)
                    flags = f if flags is None else (flags | f)
                for sim in self.engines:
                    sim.spawn_commit(flags)
        for sim in self.engines:
            sim.step(dur, report=report)
        if getattr(self, "_host_planner", False) or getattr(self, "_route_legs", False):
            # legs of route followers that missed the route book: every tile plans them, in agent
            # order, so that all books number routes alike (cs_route_resolve)
            misses = sorted(m for sim in self.engines for m in sim.route_misses())
            if misses:
                for sim in self.engines:
                    sim.route_resolve(misses)

    # SpatialIndex on a mesh (spatial_index.rs:4-14): every tile answers for the agents it owns
    def get_neighbours_in_radius_batch(self, radii, positions):
        per_tile = [sim.query_radius_batch(radii, positions, details=True) for sim in self.engines]
        return [merge_radius_answers([t[q] for t in per_tile]) for q in range(len(per_tile[0]))]

    def get_neighbours_in_radius(self, radius, position):
        return self.get_neighbours_in_radius_batch([radius], [position])[0]

    def get_nearest_neighbours_batch(self, n, positions):
        per_tile = [sim.query_knn_batch(n, positions, details=True) for sim in self.engines]
        return [merge_knn_answers([t[q] for t in per_tile], int(n)) for q in range(len(per_tile[0]))]

    def get_nearest_neighbours(self, n, position):
        return self.get_nearest_neighbours_batch(n, [position])[0]

    def read_agents(self):
        parts = [sim.read_agents() for sim in self.engines]
        out = np.concatenate(parts)
        return out[np.argsort(out["id"], kind="stable")]

    def __len__(self):
        return sum(len(sim) for sim in self.engines)


class _MeshRegistrar:
    """Lets a planner's `_register(lib, engine)` register it with a mesh: cs_register_x(engine, ...) becomes
    cs_mesh_register_x(mesh, ...)."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        if name.startswith("cs_register_") and hasattr(self._lib, "cs_mesh_register_" + name[len("cs_register_"):]):
            return getattr(self._lib, "cs_mesh_register_" + name[len("cs_register_"):])
        if name == "cs_last_error":
            return self._lib.cs_mesh_last_error
        raise AttributeError(f"{name}: not available on a mesh")


class TorchHostTransport:
    """cs_mesh_host_transport over torch.distributed (any backend that moves CPU tensors: gloo, mpi): the halo
    records, the spawn flags and the gathers of a distributed cs_mesh through a transport the HOST brings instead of
    (or beside) RCCL.  The library stages device data through pinned host memory; these callbacks only see host
    pointers.  A host in another language passes its own three functions (MPI_Sendrecv / MPI_Allreduce /
    MPI_Allgather would do)."""

    def __init__(self, dist=None, group=None):
        import ctypes as C
        import torch
        import torch.distributed as tdist
        self.dist, self.group, self.torch, self._C = dist or tdist, group, torch, C
        self.failure = None

        def view(ptr, nbytes, dtype=torch.uint8):
            raw = (C.c_ubyte * nbytes).from_address(ptr)
            return torch.frombuffer(raw, dtype=dtype)

        def guard(fn):
            def run(*args):
                try:
                    fn(*args)
                    return 0
                except Exception as err:  # (an exception must not cross the C frame)
                    self.failure = err
                    return 1
            return run

        def exchange(_user, n, peers, send_tags, recv_tags, send_host, recv_host, nbytes):
            ops = []
            for k in range(n):
                if nbytes[k] == 0:
                    continue
                ops.append(self.dist.P2POp(self.dist.isend, view(send_host[k], nbytes[k]), int(peers[k]), self.group,
                                           int(send_tags[k])))
                ops.append(self.dist.P2POp(self.dist.irecv, view(recv_host[k], nbytes[k]), int(peers[k]), self.group,
                                           int(recv_tags[k])))
            if ops:
                for req in self.dist.batch_isend_irecv(ops):
                    req.wait()

        def allreduce_max(_user, values, n):
            if n:
                t = view(C.addressof(values.contents), 4 * n, torch.int32)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

        def allgather(_user, mine, nbytes, everything):
            if nbytes:
                world = self.dist.get_world_size(self.group)
                out = list(view(everything, nbytes * world).chunk(world))
                self.dist.all_gather(out, view(mine, nbytes), group=self.group)

        self._keep = (_abi.MeshExchangeFn(guard(exchange)), _abi.MeshAllreduceMaxFn(guard(allreduce_max)),
                      _abi.MeshAllgatherFn(guard(allgather)))
        self.struct = _abi.MeshHostTransport(None, *self._keep)


class NativeTileMesh:
    """A crowd cut into tiles behind the C ABI's mesh handle (cs_mesh_*, include/crowdstep.h): layout, tile engines,
    halo buffers, exchange, spawn flags, route misses, re-cuts and merged queries all live in the library
    (csrc/cs_mesh.hip.inc); this class only converts arguments.  In-process form (every tile on one device),
    the same interface as LocalTileMesh, whose Python orchestration it replaces; with `rccl_unique_id` (bytes from
    Simulation.rccl_unique_id() on one rank, passed around by the host) and / or `host_transport` (a
    TorchHostTransport, or anything with a `.struct` of type _abi.MeshHostTransport) the distributed form: this
    rank's tile only; every call is collective then, and read_agents, len(), re-cuts and queries cover the whole crowd."""

    def __init__(self, spatial_index, tiles, halo_cells, device=0, density_per_cell=16.0, flags=0, weights=None,
                 capacity_hint=0, library=None, rank=0, n_ranks=1, rccl_unique_id=None, host_transport=None):
        import ctypes as C
        from . import _native
        from .simulation import CrowdSimError
        self._C, self._err_cls = C, CrowdSimError
        self._lib = library or _native.load()
        desc = _abi.MeshDesc()
        desc.tiles_x, desc.tiles_y = int(tiles[0]), int(tiles[1])
        desc.halo_cells, desc.flags, desc.device_ordinal = int(halo_cells), int(flags), int(device)
        desc.rank, desc.n_ranks = int(rank), int(n_ranks)
        uid = None
        if rccl_unique_id is not None:  # the distributed form: one tile per rank, halo records over RCCL
            uid = (C.c_uint8 * len(rccl_unique_id)).from_buffer_copy(bytes(rccl_unique_id))
            desc.rccl_unique_id = C.cast(uid, C.POINTER(C.c_uint8))
        self._host_transport = host_transport  # (the callbacks must outlive the mesh)
        if host_transport is not None:
            desc.host_transport = C.pointer(host_transport.struct)
        desc.density_per_cell, desc.capacity_hint = float(density_per_cell), int(capacity_hint)
        w = None
        if weights is not None:
            w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64).reshape(-1, 2))
            desc.weights_xy, desc.n_weights = w.ctypes.data_as(C.POINTER(C.c_double)), len(w)
        grid = spatial_index._desc()
        self._mesh = self._lib.cs_mesh_create(C.byref(grid), C.byref(desc))
        if not self._mesh:
            raise CrowdSimError("cs_mesh_create failed: " + self._lib.cs_mesh_last_error(None).decode())
        self._registrar = _MeshRegistrar(self._lib)
        self._handles, self._alive, self._listeners = {}, [], []
        self._host_lp_of_agent, self._host_lp_of_sink = {}, {}  # LocalPlanner::remove_agent follows the destroy events
        self.last_report = None
        self.shape = (desc.tiles_x, desc.tiles_y)

    def __del__(self):
        if getattr(self, "_mesh", None):
            self._lib.cs_mesh_destroy(self._mesh)
            self._mesh = None

    def _err(self):
        why = self._lib.cs_mesh_last_error(self._mesh).decode()
        cause = getattr(self._host_transport, "failure", None)
        for planner in self._alive:  # a host planner that raised: its exception is the cause
            if cause is None and getattr(planner, "failure", None) is not None:
                cause, planner.failure = planner.failure, None
        return self._err_cls(why + (f" ({cause!r})" if cause is not None else ""))

    def _handle(self, planner):
        key = id(planner)
        if key not in self._handles:
            handle = planner._register(self._registrar, self._mesh)
            if handle == 0xFFFFFFFF:
                raise self._err()
            self._handles[key] = handle
            self._alive.append(planner)
        return self._handles[key]

    def _dispatch(self):
        if not self._listeners and not self._host_lp_of_agent and not self._host_lp_of_sink:
            return
        buf = (_abi.Event * 4096)()
        while True:
            n = self._lib.cs_mesh_drain_events(self._mesh, buf, len(buf))
            for i in range(n):
                if buf[i].kind == _abi.CS_EVENT_SPAWNED and buf[i].source_sink in self._host_lp_of_sink:
                    self._host_lp_of_agent[int(buf[i].id)] = self._host_lp_of_sink[buf[i].source_sink]
                elif buf[i].kind == _abi.CS_EVENT_DESTROYED:
                    planner = self._host_lp_of_agent.pop(int(buf[i].id), None)
                    if planner is not None:
                        planner.remove_agent(int(buf[i].id))   # local_planner.rs:16 via lib.rs:181-184
                for listener in self._listeners:
                    if buf[i].kind == _abi.CS_EVENT_SPAWNED:
                        listener.agent_spawned(np.array([buf[i].x, buf[i].y]), int(buf[i].id))
                    elif buf[i].kind == _abi.CS_EVENT_DESTROYED:
                        listener.agent_destroyed(int(buf[i].id))
            if n < len(buf):
                break

    def add_agents(self, positions, high_level_planner, local_planner, eyesight):
        C = self._C
        pts = np.ascontiguousarray(np.asarray(positions, dtype=np.float64).reshape(-1, 2))
        ids = np.zeros(len(pts), dtype=np.uint64)
        rc = self._lib.cs_mesh_add_agents(self._mesh, pts.ctypes.data_as(C.POINTER(C.c_double)), len(pts),
                                          self._handle(high_level_planner), self._handle(local_planner), float(eyesight),
                                          ids.ctypes.data_as(C.POINTER(C.c_uint64)))
        if getattr(local_planner, "_host_code", False) and rc == 0:
            self._lib.cs_mesh_event_recording(self._mesh, 1)
            for i in ids:
                self._host_lp_of_agent[int(i)] = local_planner
        self._dispatch()
        if rc != 0:
            raise self._err()
        return [int(i) for i in ids]

    def add_source_sink(self, source_sink):
        from .simulation import source_sink_desc
        desc, keep = source_sink_desc(source_sink, self._handle)
        handle = self._lib.cs_mesh_add_source_sink(self._mesh, self._C.byref(desc))
        if handle == 0xFFFFFFFF:
            raise self._err()
        self._alive.append((source_sink, keep))
        if getattr(source_sink.local_planner, "_host_code", False):
            self._host_lp_of_sink[handle] = source_sink.local_planner
            self._lib.cs_mesh_event_recording(self._mesh, 1)
        return handle

    def remove_source_sink(self, handle):
        self._lib.cs_mesh_remove_source_sink(self._mesh, int(handle))

    def add_event_listener(self, listener):
        self._listeners.append(listener)
        self._lib.cs_mesh_event_recording(self._mesh, 1)

    def remove_agents(self, agent):
        rc = self._lib.cs_mesh_remove_agent(self._mesh, int(agent))
        self._dispatch()
        if rc != 0:
            raise self._err()

    def step(self, dur, report=True):
        rep = _abi.StepReport()
        need = report or bool(self._listeners) or bool(self._host_lp_of_agent) or bool(self._host_lp_of_sink)
        rc = self._lib.cs_mesh_step(self._mesh, float(dur), self._C.byref(rep) if need else None)
        if need:
            self.last_report = rep.as_dict()
        self._dispatch()
        if rc != 0:
            raise self._err()

    def synchronize(self):
        if self._lib.cs_mesh_synchronize(self._mesh) != 0:
            raise self._err()

    def tile(self, local_index=0):
        """The engine of a local tile as a (borrowed) Simulation: profiling, kernel statistics, snapshots."""
        from .simulation import Simulation
        engine = self._lib.cs_mesh_tile(self._mesh, int(local_index))
        if not engine:
            raise IndexError(local_index)
        return Simulation.borrowed(self._lib, engine)

    @property
    def exchange_bytes(self):
        """Bytes the local tiles send per halo exchange (fixed-capacity buffers)."""
        return int(self._lib.cs_mesh_exchange_bytes(self._mesh))

    def recut(self):
        if self._lib.cs_mesh_recut(self._mesh) != 0:
            raise self._err()
        return self.tile_counts()

    def tile_counts(self):
        n = self._lib.cs_mesh_local_tiles(self._mesh)
        out = np.zeros(n, dtype=np.uint64)
        self._lib.cs_mesh_tile_counts(self._mesh, out.ctypes.data_as(self._C.POINTER(self._C.c_uint64)))
        return out.astype(np.int64).reshape(self.shape) if n == self.shape[0] * self.shape[1] else out.astype(np.int64)

    def tile_rects(self):
        n = self._lib.cs_mesh_local_tiles(self._mesh)
        out = np.zeros((n, 4), dtype=np.uint32)
        for k in range(n):
            self._lib.cs_mesh_tile_rect(self._mesh, k, out[k].ctypes.data_as(self._C.POINTER(self._C.c_uint32)))
        return out

    def read_agents(self):
        from .simulation import AGENT_DTYPE
        self.synchronize()
        n = self._lib.cs_mesh_agent_count(self._mesh)
        buf = (_abi.AgentView * max(n, 1))()
        got = self._lib.cs_mesh_read_agents(self._mesh, buf, n)
        if got == self._C.c_size_t(-1).value:
            raise self._err()
        return np.frombuffer(buf, dtype=AGENT_DTYPE, count=got).copy()

    def __len__(self):
        return int(self._lib.cs_mesh_agent_count(self._mesh))

    def get_neighbours_in_radius_batch(self, radii, positions):
        C = self._C
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float64).reshape(-1, 2))
        n = len(pos)
        rad = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, dtype=np.float64), (n,)))
        cap = 64
        while True:
            ids, counts = np.zeros((n, cap), dtype=np.uint64), np.zeros(n, dtype=np.uint64)
            if self._lib.cs_mesh_query_radius_batch(self._mesh, n, pos.ctypes.data_as(C.POINTER(C.c_double)),
                                                    rad.ctypes.data_as(C.POINTER(C.c_double)), cap,
                                                    ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                    counts.ctypes.data_as(C.POINTER(C.c_uint64))) != 0:
                raise self._err()
            if n == 0 or counts.max() <= cap:
                return [[int(v) for v in ids[i, :int(counts[i])]] for i in range(n)]
            cap = int(counts.max())

    def get_neighbours_in_radius(self, radius, position):
        return self.get_neighbours_in_radius_batch([radius], [position])[0]

    def get_nearest_neighbours_batch(self, k, positions):
        C = self._C
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float64).reshape(-1, 2))
        n, k = len(pos), int(k)
        ids, counts = np.zeros((n, max(k, 1)), dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        if self._lib.cs_mesh_query_knn_batch(self._mesh, n, pos.ctypes.data_as(C.POINTER(C.c_double)), k,
                                             ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                             counts.ctypes.data_as(C.POINTER(C.c_uint64))) != 0:
            raise self._err()
        return [[int(v) for v in ids[i, :int(counts[i])]] for i in range(n)]

    def get_nearest_neighbours(self, k, position):
        return self.get_nearest_neighbours_batch(k, [position])[0]


def exchange_axis(dist, layout, index, bufs, axis, op_cache=None):
    """One phase of the exchange for the tile `index` = this rank: post the sends of this axis
    and the matching receives as one batch (P2P over xGMI with the nccl/RCCL backend).  With
    device buffers on nccl the P2POp list never changes; `op_cache` (a dict) keeps it."""
    if op_cache is not None and axis in op_cache:
        ops = op_cache[axis]
        if ops:
            for work in dist.batch_isend_irecv(ops):
                work.wait()
        return
    tx, ty = layout.coords(index)
    ops, staged = [], []
    for d in ((XLO, XHI) if axis == 0 else (YLO, YHI)):
        peer = layout.neighbour(tx, ty, d)
        if peer is None:
            continue
        send, recv = bufs[d]
        if send.is_cuda and dist.get_backend() == "gloo":
            # test transport (several ranks sharing one GPU): gloo moves host memory
            host_recv = recv.cpu()
            staged.append((recv, host_recv))
            send, recv = send.cpu(), host_recv
        ops.append(dist.P2POp(dist.isend, send, peer))
        ops.append(dist.P2POp(dist.irecv, recv, peer))
    if op_cache is not None and not staged:
        op_cache[axis] = ops
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    for dev_recv, host_recv in staged:
        dev_recv.copy_(host_recv)


def exchange_all(dist, layout, index, bufs, op_cache=None):
    """The one-phase exchange for the tile `index` = this rank: every send and the matching
    receive (edges and corners) in one batch."""
    if op_cache is not None and "all" in op_cache:
        ops = op_cache["all"]
        if ops:
            for work in dist.batch_isend_irecv(ops):
                work.wait()
        return
    tx, ty = layout.coords(index)
    ops, staged = [], []
    for d in ALL_DIRS:
        peer = layout.neighbour(tx, ty, d)
        if peer is None or d not in bufs:
            continue
        send, recv = bufs[d]
        if send.is_cuda and dist.get_backend() == "gloo":
            host_recv = recv.cpu()
            staged.append((recv, host_recv))
            send, recv = send.cpu(), host_recv
        ops.append(dist.P2POp(dist.isend, send, peer))
        ops.append(dist.P2POp(dist.irecv, recv, peer))
    if op_cache is not None and not staged:
        op_cache["all"] = ops
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    for dev_recv, host_recv in staged:
        dev_recv.copy_(host_recv)


class DistributedTiles(_TileBase):
    """One tile per rank (rank == tile index) under an initialised torch.distributed group."""

    def __init__(self, spatial_index, tiles, halo_cells, device, capacity_records=None,
                 density_per_cell=16.0, capacity_hint=0, flags=0, weights=None, phases=1, transport=None):
        """transport: "engine" = the C ABI's own RCCL transport (cs_halo_exchange_rccl: ncclSend /
        ncclRecv issued by the engine on its stream, what a Rust or C++ host uses), "torch" =
        torch.distributed batch_isend_irecv.  Default: "engine" on the nccl backend (CS_TILES_TRANSPORT
        overrides), "torch" otherwise (the gloo test double moves host memory)."""
        import os
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.phases = int(phases)
        self.layout = TileLayout(spatial_index, *tiles, weights=weights, min_cells=2 * int(halo_cells))
        assert self.layout.n_tiles == dist.get_world_size(), "one rank per tile"
        assert self.layout.min_tile_cells() >= 2 * int(halo_cells), "tiles thinner than two halos"
        self.index = dist.get_rank()
        dev = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(dev)  # shared by the engine and the P2P ops (see LocalTileMesh)
        stream = self.stream.cuda_stream
        self.sim = self._make_engine(spatial_index, self.layout, self.index, halo_cells, device, stream,
                                     capacity_hint, flags)
        tx, ty = self.layout.coords(self.index)
        self.spatial_index, self.halo_cells, self.bufs_device = spatial_index, int(halo_cells), dev
        self._capacity_records, self._density_per_cell = capacity_records, density_per_cell
        self.bufs = self._set_buffers()
        self._op_cache = {} if dist.get_backend() == "nccl" else None
        if transport is None:
            transport = os.environ.get("CS_TILES_TRANSPORT", "engine" if dist.get_backend() == "nccl" else "torch")
        self.transport = transport
        if transport == "engine":
            # one RCCL communicator per engine: rank 0's unique id reaches the others through the
            # process group that already exists; from then on the halo traffic needs no torch
            failure = None
            try:
                box = [self.sim.rccl_unique_id() if self.index == 0 else None]
            except Exception as err:  # librccl could not be bound on rank 0
                box, failure = [None], err
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                try:
                    self.sim.rccl_comm_init(dist.get_world_size(), self.index, box[0])
                    peers = [self.layout.neighbour(tx, ty, d) for d in ALL_DIRS]
                    self.sim.halo_set_peers([-1 if (p is None or d not in self.bufs) else p
                                             for p, d in zip(peers, ALL_DIRS)])
                except Exception as err:
                    failure = err
            # every rank must use the same transport: if the engine's failed anywhere, all fall back
            bad = torch.tensor([1 if (failure is not None or box[0] is None) else 0], dtype=torch.int32)
            if dist.get_backend() == "nccl":
                bad = bad.to(dev)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                import sys
                print(f"crowdstep: the engine's RCCL transport is not available ({failure}); using torch.distributed",
                      file=sys.stderr)
                self.transport = "torch"
        torch.cuda.synchronize(dev)

    def _set_buffers(self):
        tx, ty = self.layout.coords(self.index)
        bufs = {}
        for d in (EDGES if self.phases == 2 else ALL_DIRS):
            if self.layout.neighbour(tx, ty, d) is None:
                continue
            cap = self._capacity_records or (
                halo_capacity(self.layout, self._density_per_cell, self.halo_cells) if self.phases == 2 else
                halo_capacity_of(self.layout, tx, ty, d, self._density_per_cell, self.halo_cells))
            send, recv = self._alloc(self.torch, cap, self.bufs_device), self._alloc(self.torch, cap, self.bufs_device)
            self.sim.halo_set_buffers(d, send.data_ptr(), recv.data_ptr(), cap)
            bufs[d] = (send, recv)
        return bufs

    def recut(self):
        """Collective: new cuts at the quantiles of where the crowd stands now (LocalTileMesh.recut).
        The histograms are summed over the ranks, the exported agents gathered on every rank (host
        staged: a re-cut is rare), every rank keeps what its new rectangle owns."""
        rows = np.zeros(self.layout.rows, dtype=np.uint64)
        cols = np.zeros(self.layout.cols, dtype=np.uint64)
        self.sim.tile_histogram(rows, cols)
        both = self.torch.from_numpy(np.concatenate([rows, cols]).astype(np.int64))
        if self.dist.get_backend() == "nccl":
            both = both.to(self.bufs_device)
        self.dist.all_reduce(both)
        both = both.cpu().numpy()
        self.layout = TileLayout(self.spatial_index, self.layout.tiles_x, self.layout.tiles_y,
                                 min_cells=2 * self.halo_cells, histograms=(both[:len(rows)], both[len(rows):]))
        parts = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(parts, self.sim.tile_export())
        self.torch.cuda.synchronize(self.bufs_device)
        self.sim.tile_retile(self.layout.rect(*self.layout.coords(self.index)))
        self.bufs = self._set_buffers()
        self._op_cache = {} if self._op_cache is not None else None  # the P2P ops held the old buffers
        self.torch.cuda.synchronize(self.bufs_device)
        for rec in parts:
            self.sim.tile_import(rec)
        counts = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(counts, len(self.sim))
        return np.array(counts).reshape(self.layout.tiles_x, self.layout.tiles_y)

    def add_agents(self, positions, high_level_planner, local_planner, eyesight):
        return self.sim.add_agents(positions, high_level_planner, local_planner, eyesight)

    def add_source_sink(self, source_sink):
        self._has_sinks = True
        self._n_sinks = getattr(self, "_n_sinks", 0) + 1
        self._route_legs = getattr(self, "_route_legs", False) or _has_route_legs(source_sink)
        # (legs that may miss the route book need the host after the step: the host-side spawn path)
        self._host_planner = (getattr(self, "_host_planner", False) or not _is_data_planner(source_sink) or
                              self._route_legs)
        return self.sim.add_source_sink(source_sink)

    def add_event_listener(self, listener):
        """Events of this rank's tile (spawns of the sinks it owns, removals of agents it holds)."""
        return self.sim.add_event_listener(listener)

    def remove_source_sink(self, handle):
        self.sim.remove_source_sink(handle)

    def remove_agents(self, agent):
        """lib.rs:176-192 on every rank (collective): the owner removes, the others learn of it."""
        from .simulation import CrowdSimError
        failure = None
        try:
            found = self.sim.remove_agent_here(agent)
        except CrowdSimError as err:  # this rank's engine failed: still take part in the collective
            found, failure = False, err
        flag = self.torch.tensor([1 if found else 0, 1 if failure else 0], dtype=self.torch.int32)
        if self.dist.get_backend() == "nccl":
            flag = flag.to(self.stream.device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        if failure is not None:
            raise failure
        if int(flag[1].item()):
            raise CrowdSimError("remove_agents failed on another rank")
        if int(flag[0].item()) == 0:
            raise CrowdSimError("unknown agent id")

    def step(self, dur, report=False):
        if (self.transport == "engine" and self.phases == 1 and not getattr(self, "_host_planner", False) and
                not self.sim.host_events_needed):
            # nothing needs the host between the phases: the whole step is one call into the engine
            self.sim.tile_step_rccl(dur, report=report)
            return
        with self.torch.cuda.stream(self.stream):
            if self.phases == 2:
                for axis in (0, 1):
                    self.sim.halo_pack(axis)
                    if self.transport == "engine":
                        self.sim.halo_exchange_rccl(axis)
                    else:
                        exchange_axis(self.dist, self.layout, self.index, self.bufs, axis, self._op_cache)
                    self.sim.halo_unpack(axis)
            else:
                self.sim.halo_pack_all()
                if self.transport == "engine":
                    self.sim.halo_exchange_rccl(-1)
                else:
                    exchange_all(self.dist, self.layout, self.index, self.bufs, self._op_cache)
                self.sim.halo_unpack_all()
        if getattr(self, "_has_sinks", False):
            # Ids follow the global sink order: OR the per-tile spawn flags.  Whether a rank probes
            # and commits through its host (listeners, host planners, a report) or keeps the flags
            # on the device is that rank's own business; what the ranks share is ONE all-reduce of
            # one int32 per sink slot, issued from this single place whichever path a rank takes,
            # so the ranks cannot fall out of step with each other.
            host_path = report or self._host_planner or self.sim.host_events_needed
            n = self.sim.source_sink_slots
            with self.torch.cuda.stream(self.stream):
                if getattr(self, "_flags_dev", None) is None or self._flags_dev.numel() < n:
                    self._flags_dev = self.torch.zeros(max(n, 1), dtype=self.torch.int32,
                                                       device=self.stream.device)
                flags = self._flags_dev[:max(n, 1)]
                if host_path:
                    mine = self.torch.from_numpy(self.sim.spawn_probe(dur).astype(np.int32))
                    flags[:len(mine)].copy_(mine)
                else:
                    self.sim.spawn_probe_dev(dur, flags.data_ptr(), n)
                self._allreduce_max(flags)
                if host_path:
                    self.sim.spawn_commit(flags[:n].cpu().numpy().astype(np.uint8))
                else:
                    self.sim.spawn_commit_dev(flags.data_ptr(), n)
        self.sim.step(dur, report=report)
        if getattr(self, "_route_legs", False):
            # legs of route followers that missed the route book on ANY tile: all ranks plan them in
            # agent order, so that every book numbers routes alike (cs_route_resolve)
            mine = self.sim.route_misses()
            count = self.torch.tensor([len(mine)], dtype=self.torch.int32)
            if self.dist.get_backend() == "nccl":
                count = count.to(self.stream.device)
            self.dist.all_reduce(count, op=self.dist.ReduceOp.SUM)
            if int(count.item()):
                parts = [None] * self.dist.get_world_size()
                self.dist.all_gather_object(parts, mine)
                self.sim.route_resolve(sorted(m for part in parts for m in part))

    # SpatialIndex on a mesh, collective: every rank asks the same queries and gets the full answers
    def get_neighbours_in_radius_batch(self, radii, positions):
        mine = self.sim.query_radius_batch(radii, positions, details=True)
        parts = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(parts, mine)
        return [merge_radius_answers([t[q] for t in parts]) for q in range(len(mine))]

    def get_neighbours_in_radius(self, radius, position):
        return self.get_neighbours_in_radius_batch([radius], [position])[0]

    def get_nearest_neighbours_batch(self, n, positions):
        mine = self.sim.query_knn_batch(n, positions, details=True)
        parts = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(parts, mine)
        return [merge_knn_answers([t[q] for t in parts], int(n)) for q in range(len(mine))]

    def get_nearest_neighbours(self, n, position):
        return self.get_nearest_neighbours_batch(n, [position])[0]

    def _allreduce_max(self, t):
        """MAX over the ranks of a device tensor, on the engine's stream."""
        if self.transport == "engine":
            self.sim.allreduce_max_rccl(t.data_ptr(), t.numel())
        elif self.dist.get_backend() == "nccl":
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        else:  # test transport (ranks sharing one GPU): gloo moves host memory
            host = t.cpu()
            self.dist.all_reduce(host, op=self.dist.ReduceOp.MAX)
            t.copy_(host)

    def read_agents(self):
        return self.sim.read_agents()

    def __len__(self):
        return len(self.sim)
