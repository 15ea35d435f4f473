"""Host-side mirror of the reference's trait surface for the `Simulation::step` path.

Same names and argument meaning as rmf_crowdsim (paths relative to
rmf_crowdsim/src in the reference tree):

    Simulation            lib.rs:69-383      new / add_agents / add_source_sink /
                                             remove_source_sink / add_event_listener /
                                             remove_agents / step / agents
    EventListener         lib.rs:22-33
    Agent                 lib.rs:46-65
    HighLevelPlanner      highlevel_planners/highlevel_planners.rs:8-16
    LocalPlanner          local_planners/local_planner.rs:7-18
    Zanlungo / NoLocalPlan  local_planners/zanlungo.rs:31-48, no_local_plan.rs:7-18
    LocationHash2D        spatial_index/location_hash_2d.rs:33-51
    SourceSink / CrowdGenerator / MonotonicCrowd   source_sink/source_sink.rs:30-101

Everything forwards to the C ABI (include/crowdstep.h) of the HIP engine; Rust
`Result<_, String>` becomes `CrowdSimError(message)`.
"""
import ctypes as C
import datetime
from dataclasses import dataclass

import numpy as np

from . import _abi, _native


class CrowdSimError(RuntimeError):
    """`Err(String)` of the reference API ("Index out of bounds", ...)."""


@dataclass
class Agent:
    """pub struct Agent, lib.rs:46-65 (the fields `step` maintains)."""
    agent_id: int
    position: np.ndarray
    velocity: np.ndarray
    next_waypoint: int
    eyesight_range: float
    orientation: float = 0.0   # never written after creation (lib.rs:138)
    angular_vel: float = 0.0   # never written after creation (lib.rs:141)
    preferred_vel: object = None  # set on the clone a LocalPlanner sees (lib.rs:271); (0, 0) on its neighbours


# ---- spatial index ---------------------------------------------------------
class SpatialIndex:
    """trait SpatialIndex, spatial_index.rs:4-14: what `Simulation<T: SpatialIndex>` (lib.rs:69) is generic over.

    On this backend the index IS the neighbour kernel: the engine keeps the agents in the cell order of a
    `LocationHash2D` and answers `get_neighbours_in_radius` / `get_nearest_neighbours` / `add_or_update` /
    `remove_agent` itself (`Simulation.get_neighbours_in_radius`, `.get_nearest_neighbours`, the step's re-binning,
    `.remove_agents`).  An index therefore has to describe itself as such a grid: `device_form()` returns the
    `LocationHash2D` it is equivalent to (the one provided method this mirror adds to the trait, like the planners'), or
    None, in which case `Simulation(index)` refuses it with an error that says so.  There is no host-side slow path
    for a foreign index (unlike the planners: a per-agent host query per step would put the hot path on the CPU)."""

    def add_or_update(self, index, position):           # spatial_index.rs:6
        raise NotImplementedError

    def get_nearest_neighbours(self, n, position):       # spatial_index.rs:8
        raise NotImplementedError

    def get_neighbours_in_radius(self, radius, position):  # spatial_index.rs:10
        raise NotImplementedError

    def remove_agent(self, index):                       # spatial_index.rs:12
        raise NotImplementedError

    def device_form(self):
        return None


class LocationHash2D(SpatialIndex):
    """LocationHash2D::new(width, height, cell_size, offset); location_hash_2d.rs:33."""

    def device_form(self):
        return self

    def __init__(self, width, height, cell_size, offset):
        self.width = float(width)
        self.height = float(height)
        self.cell_size = float(cell_size)
        self.offset = (float(offset[0]), float(offset[1]))

    def _desc(self):
        return _abi.GridDesc(self.width, self.height, self.cell_size, *self.offset)


# ---- local planners --------------------------------------------------------
class LocalPlanner:
    """trait LocalPlanner, local_planner.rs:7-18.

    The two planners the reference ships (Zanlungo, NoLocalPlan) run on the device; they are passed
    as data.  Any other subclass is host code: override `get_desired_velocity`, and the engine
    evaluates it through the batched callback each step (cs_register_lp_callback: the slow path of
    SURVEY.md section 8b), with the agent and its neighbours as they were at the start of the step.
    """

    def get_desired_velocity(self, agent, nearby_agents, recommended_velocity):
        """-> (vx, vy).  `agent.preferred_vel` is the recommended velocity too (lib.rs:271); the
        neighbours' is (0, 0), as in the reference (lib.rs:140)."""
        raise NotImplementedError

    def add_agent(self, agent_id):       # local_planner.rs:14 (the reference never calls it: lib.rs:127-131 only
        pass                             # stores the planner per agent)

    def remove_agent(self, agent_id):    # local_planner.rs:16, called by remove_agents (lib.rs:181-184)
        pass

    _host_code = True  # the facade tells such a planner when one of its agents is removed

    def _register(self, lib, engine):
        if type(self).get_desired_velocity is LocalPlanner.get_desired_velocity:
            raise CrowdSimError("a LocalPlanner must be Zanlungo, NoLocalPlan or override get_desired_velocity")

        def batch(_user, n, agents, recommended, nb_begin, neighbours, out):
            def view(r, preferred):
                a = Agent(int(r.agent_id), np.array([r.x, r.y]), np.array([r.vx, r.vy]), int(r.next_waypoint),
                          float(r.eyesight_range))
                a.preferred_vel = preferred
                return a
            try:
                for k in range(n):
                    rec = np.array([recommended[2 * k], recommended[2 * k + 1]])
                    me = view(agents[k], rec)
                    nearby = [view(neighbours[q], np.zeros(2)) for q in range(nb_begin[k], nb_begin[k + 1])]
                    vx, vy = self.get_desired_velocity(me, nearby, rec)
                    out[2 * k], out[2 * k + 1] = float(vx), float(vy)
            except Exception as err:  # noqa: BLE001 (an exception must not cross the C frame: the step fails instead)
                self.failure = err
                return 1
            return 0

        fn = _abi.LpBatchFn(batch)
        # one thunk per engine this planner is registered with; the engines hold the raw pointer
        self._keepalive = getattr(self, "_keepalive", []) + [fn]
        handle = lib.cs_register_lp_callback(engine, fn, None)
        if handle == 0xFFFFFFFF:
            raise CrowdSimError(lib.cs_last_error(engine).decode())
        return handle


class NoLocalPlan(LocalPlanner):
    """no_local_plan.rs:7-18: returns the recommended velocity unchanged."""
    _host_code = False

    def _register(self, lib, engine):
        return lib.cs_register_no_local_plan(engine)


class Zanlungo(LocalPlanner):
    """Zanlungo::new(agent_scale, obstacle_scale, reaction_time, force_distance,
    agent_mass, agent_radius); zanlungo.rs:31-48."""

    _host_code = False

    def __init__(self, agent_scale, obstacle_scale, reaction_time, force_distance, agent_mass,
                 agent_radius):
        self.params = _abi.ZanlungoParams(agent_scale, obstacle_scale, reaction_time,
                                          force_distance, agent_mass, agent_radius)

    def _register(self, lib, engine):
        return lib.cs_register_zanlungo(engine, C.byref(self.params))


# ---- high-level planners ---------------------------------------------------
class HighLevelPlanner:
    """trait HighLevelPlanner, highlevel_planners.rs:8-16.

    Subclass and override `get_desired_velocity` for a host planner (evaluated
    through the batched callback each step: the slow path), or use the data
    planners below, which the device evaluates itself.
    """

    def get_desired_velocity(self, agent, time):
        """-> (vx, vy) or None"""
        return None

    def set_target(self, agent, point, tolerance):
        pass

    def remove_agent_id(self, agent_id):
        pass

    # -- ABI plumbing --
    def _desc(self):
        keep = []

        def velocity(_user, n, ids, pos, vel, time_s, out, some):
            for i in range(n):
                agent = Agent(int(ids[i]), np.array([pos[2 * i], pos[2 * i + 1]]),
                              np.array([vel[2 * i], vel[2 * i + 1]]), 0, 0.0)
                res = self.get_desired_velocity(agent, datetime.timedelta(seconds=time_s))
                if res is None:
                    some[i] = 0
                else:
                    some[i] = 1
                    out[2 * i], out[2 * i + 1] = float(res[0]), float(res[1])

        def set_target(_user, aid, px, py, tx, ty, tolx, toly):
            agent = Agent(int(aid), np.array([px, py]), np.zeros(2), 0, 0.0)
            self.set_target(agent, np.array([tx, ty]), np.array([tolx, toly]))

        def remove(_user, aid):
            self.remove_agent_id(int(aid))

        fv, fs, fr = _abi.HlpVelocityFn(velocity), _abi.HlpSetTargetFn(set_target), \
            _abi.HlpRemoveFn(remove)
        keep += [fv, fs, fr]
        return _abi.HlpDesc(_abi.CS_HLP_CALLBACK, 0.0, 0.0, fv, fs, fr, None, _abi.RoutePlanFn(),
                            0.0, 0.0, 0.0), keep

    def _register(self, lib, engine):
        desc, keep = self._desc()
        # one set of callback thunks per engine this planner is registered with (a tile mesh
        # registers it with every tile); the engines hold the raw pointers
        self._keepalive = getattr(self, "_keepalive", []) + list(keep)
        return lib.cs_register_hlp(engine, C.byref(desc))


class _DataPlan(HighLevelPlanner):
    _kind = _abi.CS_HLP_NONE

    def __init__(self, default_vel=(0.0, 0.0)):
        self.default_vel = (float(default_vel[0]), float(default_vel[1]))

    def _desc(self):
        return _abi.HlpDesc(self._kind, self.default_vel[0], self.default_vel[1],
                            _abi.HlpVelocityFn(), _abi.HlpSetTargetFn(), _abi.HlpRemoveFn(),
                            None, _abi.RoutePlanFn(), 0.0, 0.0, 0.0), []


class NoHighLevelPlan(_DataPlan):
    """get_desired_velocity -> None for every agent (lib.rs:263-273 leaves vel = 0)."""
    _kind = _abi.CS_HLP_NONE


class StubHighLevelPlan(_DataPlan):
    """The reference tests' stub: Some(default_vel) for every agent (lib.rs:391-420)."""
    _kind = _abi.CS_HLP_CONSTANT

    def get_desired_velocity(self, agent, time):
        return self.default_vel


class IdParityHighLevelPlan(_DataPlan):
    """The visualiser's stub: even ids get -default_vel, odd ids +default_vel
    (rmf_crowdsim_viz/src/main.rs:20-30)."""
    _kind = _abi.CS_HLP_ID_PARITY

    def get_desired_velocity(self, agent, time):
        s = -1.0 if agent.agent_id % 2 == 0 else 1.0
        return (s * self.default_vel[0], s * self.default_vel[1])


class RouteFollower(HighLevelPlanner):
    """The follower half of RMFPlanner (rmf/mod.rs:195-242) with the route search left to the
    host: `plan_route(start, goal)` returns the waypoints of a route (the goal last) or None,
    the way RMFPlanner::plan_route does with A* over its visibility graph (rmf/mod.rs:160-192).
    It is called once per set_target whose (start, goal) SpatialHash pair is new
    (route_plans_by_location, rmf/mod.rs:217-236); following the route, that is
    get_desired_velocity (unit vector to the current waypoint, next waypoint inside 0.1,
    rmf/mod.rs:197-215), runs on the device for every agent every step.
    """

    def __init__(self, plan_route, scale=1.0, arrive=0.1, speed=1.0):
        self.plan_route = plan_route
        self.scale, self.arrive, self.speed = float(scale), float(arrive), float(speed)

    def _desc(self):
        def plan(_user, sx, sy, gx, gy, out, cap):
            pts = self.plan_route((sx, sy), (gx, gy))
            if not pts:
                return 0
            n = min(len(pts), int(cap))
            for k in range(n):
                out[2 * k], out[2 * k + 1] = float(pts[k][0]), float(pts[k][1])
            return n

        fp = _abi.RoutePlanFn(plan)
        return _abi.HlpDesc(_abi.CS_HLP_ROUTE, 0.0, 0.0, _abi.HlpVelocityFn(), _abi.HlpSetTargetFn(),
                            _abi.HlpRemoveFn(), None, fp, self.scale, self.arrive, self.speed), [fp]


# ---- source / sink ---------------------------------------------------------
class CrowdGenerator:
    """trait CrowdGenerator, source_sink.rs:30-33."""

    def get_number_to_spawn(self, time_elapsed):
        return 0

    def _fill(self, desc):
        def gen(_user, dt):
            return int(self.get_number_to_spawn(datetime.timedelta(seconds=dt)))
        fn = _abi.GeneratorFn(gen)
        desc.generator_kind = _abi.CS_GEN_CALLBACK
        desc.generator = fn
        return [fn]


class MonotonicCrowd(CrowdGenerator):
    """MonotonicCrowd::new(rate): round(dt * rate) per step; source_sink.rs:85-101."""

    def __init__(self, rate):
        self.rate = float(rate)

    def get_number_to_spawn(self, time_elapsed):
        v = time_elapsed.total_seconds() * self.rate
        return max(0, int(np.floor(abs(v) + 0.5) * np.sign(v)))

    def _fill(self, desc):
        desc.generator_kind = _abi.CS_GEN_MONOTONIC
        desc.rate = self.rate
        return []


class PoissonCrowd(CrowdGenerator):
    """PoissonCrowd::new(rate), source_sink.rs:63-82: Poisson(dt * rate) drawn from an UNSEEDED generator on every
    call (the reference's `thread_rng`), so two runs differ, as they do in the reference; asked on the host through the
    CrowdGenerator callback (a tile mesh refuses it: its tiles must draw alike).  SeededPoissonCrowd is the
    reproducible form."""

    def __init__(self, rate):
        self.rate = float(rate)
        self._rng = np.random.default_rng()  # OS entropy: unseeded like thread_rng

    def get_number_to_spawn(self, time_elapsed):
        rt = time_elapsed.total_seconds() * self.rate
        if not rt > 0.0:  # statrs' Poisson::new(lambda <= 0 or NaN) is an Err that the reference unwraps: a panic
            raise ValueError("PoissonCrowd: dt * rate must be positive (Poisson::new(rt).unwrap(), source_sink.rs:79)")
        return int(self._rng.poisson(rt))


class SeededPoissonCrowd(CrowdGenerator):
    """Seeded replacement for PoissonCrowd (source_sink.rs:63-82, whose
    thread_rng cannot be seeded): Poisson(dt * rate) from a counter-based
    generator keyed by (seed, step index)."""

    def __init__(self, rate, seed):
        self.rate = float(rate)
        self.seed = int(seed)

    def _fill(self, desc):
        desc.generator_kind = _abi.CS_GEN_POISSON_SEEDED
        desc.rate = self.rate
        desc.seed = self.seed
        return []


@dataclass
class SourceSink:
    """struct SourceSink, source_sink.rs:36-60."""
    source: tuple
    radius_sink: float
    crowd_generator: CrowdGenerator
    high_level_planner: HighLevelPlanner
    local_planner: LocalPlanner
    waypoints: list
    loop_forever: bool
    agent_eyesight_range: float


# ---- listeners -------------------------------------------------------------
class EventListener:
    """trait EventListener, lib.rs:22-33."""

    def agent_spawned(self, position, agent):
        pass

    def agent_destroyed(self, agent):
        pass

    def waypoint_reached(self, position, agent):
        """Declared by the reference, never called (lib.rs:32)."""


# ---- the simulation --------------------------------------------------------
def source_sink_desc(source_sink, handle_of):
    """cs_source_sink_desc of a SourceSink (source_sink.rs:36-60); handle_of(planner) -> its handle with the engine
    (or mesh) the sink is being added to.  Returns (desc, what must stay alive while the sink does)."""
    wps = np.ascontiguousarray(np.asarray(source_sink.waypoints, dtype=np.float64).reshape(-1, 2))
    desc = _abi.SourceSinkDesc()
    desc.source_x, desc.source_y = float(source_sink.source[0]), float(source_sink.source[1])
    desc.radius_sink = float(source_sink.radius_sink)
    keep = [source_sink.crowd_generator._fill(desc), wps]
    desc.hlp = handle_of(source_sink.high_level_planner)
    desc.lp = handle_of(source_sink.local_planner)
    desc.waypoints_xy = wps.ctypes.data_as(C.POINTER(C.c_double))
    desc.n_waypoints = wps.shape[0]
    desc.loop_forever = 1 if source_sink.loop_forever else 0
    desc.agent_eyesight_range = float(source_sink.agent_eyesight_range)
    return desc, keep


AGENT_DTYPE = np.dtype([("id", "<u8"), ("x", "<f8"), ("y", "<f8"), ("vx", "<f8"), ("vy", "<f8"),
                        ("next_waypoint", "<u8"), ("eyesight_range", "<f8")])


class Simulation:
    """Simulation<LocationHash2D>, lib.rs:69-383, on one MI355X."""

    def __init__(self, spatial_index, device=0, flags=_abi.CS_CFG_DEFAULT, capacity_hint=0,
                 stream=None, tile=None, halo_cells=0):
        self._lib = self._load_library()
        form = spatial_index.device_form() if hasattr(spatial_index, "device_form") else None
        if form is None or not hasattr(form, "_desc"):
            raise CrowdSimError(
                "Simulation<T: SpatialIndex>: on this backend the spatial index is the neighbour kernel itself; "
                f"{type(spatial_index).__name__} does not describe itself as a uniform grid (device_form() -> LocationHash2D)")
        spatial_index = form
        grid = spatial_index._desc()
        cfg = _abi.DeviceCfg(int(device), int(flags), 0, 0, 0, 0, 0, 0, int(capacity_hint),
                             C.c_void_p(stream) if stream else None)
        if tile is not None:
            cfg.tile_cx0, cfg.tile_cx1, cfg.tile_cy0, cfg.tile_cy1 = [int(t) for t in tile]
            cfg.halo_cells = int(halo_cells)
        self.spatial_index = spatial_index
        self._engine = self._lib.cs_create(C.byref(grid), C.byref(cfg))
        if not self._engine:
            raise CrowdSimError("cs_create failed: " + self._lib.cs_last_error(None).decode())
        self._lib.cs_event_recording(self._engine, 0)  # no listeners yet (lib.rs:88)
        self._planner_handles = {}
        self._planners_alive = []
        self._host_lps = False  # some LocalPlanner is host code: its agents are tracked for remove_agent
        self._host_lp_of_agent = {}
        self._host_lp_of_sink = {}
        self._listeners = {}
        self._next_listener = 0
        self._source_sinks = {}
        self._agents_cache = None
        self.last_report = None

    def _load_library(self):
        return _native.load()

    @classmethod
    def borrowed(cls, lib, engine, spatial_index=None):
        """A view of an engine somebody else owns (a tile of a mesh, cs_mesh_tile): profiling, kernel statistics,
        snapshots, queries of that tile.  Closing the view leaves the engine alone."""
        self = cls.__new__(cls)
        self._lib, self._engine, self._borrowed = lib, engine, True
        self.spatial_index = spatial_index
        self._planner_handles, self._planners_alive = {}, []
        self._host_lps, self._host_lp_of_agent, self._host_lp_of_sink = False, {}, {}
        self._listeners, self._next_listener, self._source_sinks = {}, 0, {}
        self._agents_cache, self.last_report = None, None
        return self

    def close(self):
        if getattr(self, "_engine", None):
            if not getattr(self, "_borrowed", False):
                self._lib.cs_destroy(self._engine)
            self._engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers --
    @property
    def backend(self):
        return self._lib.cs_backend_name(self._engine).decode()

    def _err(self):
        why = self._lib.cs_last_error(self._engine).decode()
        for planner in self._planners_alive:  # a host planner that raised: its exception is the cause
            cause = getattr(planner, "failure", None)
            if cause is not None:
                planner.failure = None
                err = CrowdSimError(f"{why} ({cause!r})")
                err.__cause__ = cause
                return err
        return CrowdSimError(why)

    def _handle(self, planner):
        key = id(planner)
        if key not in self._planner_handles:
            handle = planner._register(self._lib, self._engine)
            if handle == 0xFFFFFFFF:  # the engine refused the planner (cs_last_error says why)
                raise self._err()
            self._planner_handles[key] = handle
            self._planners_alive.append(planner)
        return self._planner_handles[key]

    def _dispatch_events(self):
        buf = (_abi.Event * 4096)()
        while True:
            n = self._lib.cs_drain_events(self._engine, buf, len(buf))
            for i in range(n):
                ev = buf[i]
                if self._host_lps:  # LocalPlanner::remove_agent for agents of host planners (lib.rs:181-184)
                    if ev.kind == _abi.CS_EVENT_SPAWNED and ev.source_sink in self._host_lp_of_sink:
                        self._host_lp_of_agent[int(ev.id)] = self._host_lp_of_sink[ev.source_sink]
                    elif ev.kind == _abi.CS_EVENT_DESTROYED:
                        planner = self._host_lp_of_agent.pop(int(ev.id), None)
                        if planner is not None:
                            planner.remove_agent(int(ev.id))
                for listener in self._listeners.values():
                    if ev.kind == _abi.CS_EVENT_SPAWNED:
                        listener.agent_spawned(np.array([ev.x, ev.y]), int(ev.id))
                    elif ev.kind == _abi.CS_EVENT_DESTROYED:
                        listener.agent_destroyed(int(ev.id))
            if n < len(buf):
                break

    # -- reference API --
    def add_agents(self, spawn_positions, high_level_planner, local_planner,
                   agent_eyesight_range):
        """lib.rs:119-156 -> list of agent ids (sequential from the last allocated id)."""
        pts = np.ascontiguousarray(np.asarray(spawn_positions, dtype=np.float64).reshape(-1, 2))
        n = pts.shape[0]
        ids = np.zeros(n, dtype=np.uint64)
        rc = self._lib.cs_add_agents(
            self._engine, pts.ctypes.data_as(C.POINTER(C.c_double)), n,
            self._handle(high_level_planner), self._handle(local_planner),
            float(agent_eyesight_range), ids.ctypes.data_as(C.POINTER(C.c_uint64)))
        self._agents_cache = None
        if getattr(local_planner, "_host_code", False):
            self._host_lps = True
            self._lib.cs_event_recording(self._engine, 1)
            for i in ids[:n if rc == 0 else 0]:
                self._host_lp_of_agent[int(i)] = local_planner
        self._dispatch_events()
        if rc != 0:
            raise self._err()
        return [int(i) for i in ids]

    def add_source_sink(self, source_sink):
        """lib.rs:159-161 -> handle"""
        desc, keep = source_sink_desc(source_sink, self._handle)
        handle = self._lib.cs_add_source_sink(self._engine, C.byref(desc))
        if handle == 0xFFFFFFFF:
            raise self._err()
        self._source_sinks[handle] = (source_sink, keep)
        if getattr(source_sink.local_planner, "_host_code", False):
            self._host_lps = True
            self._host_lp_of_sink[handle] = source_sink.local_planner
            self._lib.cs_event_recording(self._engine, 1)
        return handle

    def remove_source_sink(self, handle):
        """lib.rs:164-168"""
        self._lib.cs_remove_source_sink(self._engine, int(handle))
        self._source_sinks.pop(handle, None)

    def add_event_listener(self, event_listener):
        """lib.rs:171-173 -> handle"""
        handle = self._next_listener
        self._next_listener += 1
        self._listeners[handle] = event_listener
        self._lib.cs_event_recording(self._engine, 1)
        return handle

    def remove_agents(self, agent):
        """lib.rs:176-192 (an unknown id raises instead of panicking)."""
        rc = self._lib.cs_remove_agent(self._engine, int(agent))
        self._agents_cache = None
        self._dispatch_events()
        if rc != 0:
            raise self._err()

    def step(self, dur, report=True):
        """lib.rs:195-383.  `dur`: seconds (float) or datetime.timedelta."""
        dt = dur.total_seconds() if isinstance(dur, datetime.timedelta) else float(dur)
        rep = _abi.StepReport()
        need_report = report or bool(self._listeners) or self._host_lps
        rc = self._lib.cs_step(self._engine, dt, C.byref(rep) if need_report else None)
        self._agents_cache = None
        if need_report:
            self.last_report = rep.as_dict()
        if self._listeners or self._host_lps or rc != 0:
            self._dispatch_events()
        if rc != 0:
            raise self._err()

    def synchronize(self):
        if self._lib.cs_synchronize(self._engine) != 0:
            raise self._err()

    # -- observation --
    def read_agents(self):
        """Structured array (id, x, y, vx, vy, next_waypoint, eyesight_range), ascending id.
        Steps taken without a report return before the device has finished: an Err of such a
        step ("Index out of bounds") is raised here, by the first call that waits for it."""
        self.synchronize()
        n = self._lib.cs_agent_count(self._engine)
        buf = (_abi.AgentView * max(n, 1))()
        got = self._lib.cs_read_agents(self._engine, buf, n)
        return np.frombuffer(buf, dtype=AGENT_DTYPE, count=got).copy()

    @property
    def agents(self):
        """`pub agents: HashMap<AgentId, Agent>` (lib.rs:71), read back lazily."""
        if self._agents_cache is None:
            arr = self.read_agents()
            self._agents_cache = {
                int(r["id"]): Agent(int(r["id"]), np.array([r["x"], r["y"]]),
                                    np.array([r["vx"], r["vy"]]), int(r["next_waypoint"]),
                                    float(r["eyesight_range"]))
                for r in arr}
        return self._agents_cache

    def __len__(self):
        return int(self._lib.cs_agent_count(self._engine))

    def get_neighbours_in_radius(self, radius, position):
        """SpatialIndex::get_neighbours_in_radius, location_hash_2d.rs:240-258."""
        cap = 256
        while True:
            out = np.zeros(cap, dtype=np.uint64)
            n = self._lib.cs_query_radius(self._engine, float(radius), float(position[0]),
                                          float(position[1]),
                                          out.ctypes.data_as(C.POINTER(C.c_uint64)), cap)
            if n <= cap:
                return [int(i) for i in out[:n]]
            cap = int(n)

    def get_nearest_neighbours(self, n, position):
        """SpatialIndex::get_nearest_neighbours, location_hash_2d.rs:151-238."""
        out = np.zeros(max(int(n), 1) + len(self), dtype=np.uint64)
        got = self._lib.cs_query_knn(self._engine, int(n), float(position[0]), float(position[1]),
                                     out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return [int(i) for i in out[:got]]

    def query_radius_batch(self, radii, positions, details=False):
        """n radius queries in one launch (cs_query_radius_batch).  -> list of id lists in the order of
        get_neighbours_in_radius; with details=True a list of (ids, squared distances, global cells)
        arrays per query (what a tile mesh needs to merge the answers of its tiles)."""
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float64).reshape(-1, 2))
        n = len(pos)
        rad = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, dtype=np.float64), (n,)))
        cap = 64
        while True:
            ids = np.zeros((n, cap), dtype=np.uint64)
            counts = np.zeros(n, dtype=np.uint64)
            d2 = np.zeros((n, cap), dtype=np.float32)
            cells = np.zeros((n, cap), dtype=np.uint32)
            rc = self._lib.cs_query_radius_batch(
                self._engine, n, pos.ctypes.data_as(C.POINTER(C.c_double)), rad.ctypes.data_as(C.POINTER(C.c_double)),
                cap, ids.ctypes.data_as(C.POINTER(C.c_uint64)), counts.ctypes.data_as(C.POINTER(C.c_uint64)),
                d2.ctypes.data_as(C.POINTER(C.c_float)), cells.ctypes.data_as(C.POINTER(C.c_uint32)))
            if rc != 0:
                raise self._err()
            if n == 0 or counts.max() <= cap:
                break
            cap = int(counts.max())
        if details:
            return [(ids[i, :int(counts[i])].copy(), d2[i, :int(counts[i])].copy(), cells[i, :int(counts[i])].copy())
                    for i in range(n)]
        return [[int(v) for v in ids[i, :int(counts[i])]] for i in range(n)]

    def query_knn_batch(self, k, positions, details=False):
        """The k nearest agents of n points in one call (cs_query_knn_batch), nearest first, ties by id."""
        pos = np.ascontiguousarray(np.asarray(positions, dtype=np.float64).reshape(-1, 2))
        n, k = len(pos), int(k)
        ids = np.zeros((n, max(k, 1)), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint64)
        d2 = np.zeros((n, max(k, 1)), dtype=np.float32)
        rc = self._lib.cs_query_knn_batch(self._engine, n, pos.ctypes.data_as(C.POINTER(C.c_double)), k,
                                          ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          counts.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          d2.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != 0:
            raise self._err()
        if details:
            return [(ids[i, :int(counts[i])].copy(), d2[i, :int(counts[i])].copy()) for i in range(n)]
        return [[int(v) for v in ids[i, :int(counts[i])]] for i in range(n)]

    # -- streaming view for renderers (lib.rs:71 read every frame, main.rs:112-128) --
    def request_snapshot(self):
        """Queue a copy of the live agents into pinned host memory behind the steps queued so
        far; returns at once, the next step overlaps the transfer."""
        if self._lib.cs_snapshot_request(self._engine) != 0:
            raise self._err()

    def snapshot(self, wait=True):
        """The most recently requested snapshot as a structured array view (x, y, vx, vy, id,
        next_waypoint; unordered) and the number of steps it was taken after, or None when it is
        not complete yet (wait=False) or nothing was requested.  The view is valid until the
        second request_snapshot() from now."""
        out = C.POINTER(_abi.SnapshotRecord)()
        n, step = C.c_size_t(0), C.c_uint64(0)
        rc = self._lib.cs_snapshot_acquire(self._engine, 1 if wait else 0, C.byref(out), C.byref(n),
                                           C.byref(step))
        if rc in (1, 2):
            return None
        if rc != 0:
            raise self._err()
        if n.value == 0:
            return np.zeros(0, dtype=_abi.SNAPSHOT_DTYPE), int(step.value)
        buf = (C.c_char * (n.value * C.sizeof(_abi.SnapshotRecord))).from_address(C.addressof(out.contents))
        return np.frombuffer(buf, dtype=_abi.SNAPSHOT_DTYPE), int(step.value)

    def spawn_probe_dev(self, dur, flags_ptr, n):
        """Tile engines, no host wait: flags (one int32 per sink) go to device memory at flags_ptr."""
        dt = dur.total_seconds() if isinstance(dur, datetime.timedelta) else float(dur)
        if self._lib.cs_spawn_probe_dev(self._engine, dt, C.c_void_p(flags_ptr), int(n)) != 0:
            raise self._err()

    def spawn_commit_dev(self, flags_ptr, n):
        if self._lib.cs_spawn_commit_dev(self._engine, C.c_void_p(flags_ptr), int(n)) != 0:
            raise self._err()

    @property
    def host_events_needed(self):
        """True when spawn / waypoint / destroy events must reach the host (listeners, host local planners)."""
        return bool(self._listeners) or self._host_lps

    # -- tiles (multi-GPU): halo hooks, see tiles.py --
    def halo_set_buffers(self, direction, send_ptr, recv_ptr, capacity_records):
        if self._lib.cs_halo_set_buffers(self._engine, int(direction), C.c_void_p(send_ptr),
                                         C.c_void_p(recv_ptr), int(capacity_records)) != 0:
            raise self._err()

    def halo_pack(self, axis):
        if self._lib.cs_halo_pack(self._engine, int(axis)) != 0:
            raise self._err()

    def halo_unpack(self, axis):
        if self._lib.cs_halo_unpack(self._engine, int(axis)) != 0:
            raise self._err()

    def halo_pack_all(self):
        if self._lib.cs_halo_pack_all(self._engine) != 0:
            raise self._err()

    def halo_unpack_all(self):
        if self._lib.cs_halo_unpack_all(self._engine) != 0:
            raise self._err()

    # -- tiles: re-cutting a running mesh (cs_tile_histogram / _export / _retile / _import) --
    def tile_histogram(self, rows, cols):
        """Adds this tile's owned agents per global x-row / y-column into the uint64 arrays."""
        if self._lib.cs_tile_histogram(self._engine, rows.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       cols.ctypes.data_as(C.POINTER(C.c_uint64))) != 0:
            raise self._err()

    def tile_export(self):
        """Every owned agent as a halo record: a (n, CS_HALO_RECORD_BYTES) uint8 array."""
        n = self._lib.cs_tile_export(self._engine, None, 0)
        if n == C.c_size_t(-1).value:
            raise self._err()
        buf = np.zeros((max(n, 1), _abi.CS_HALO_RECORD_BYTES), dtype=np.uint8)
        got = self._lib.cs_tile_export(self._engine, buf.ctypes.data_as(C.c_void_p), n)
        if got == C.c_size_t(-1).value:
            raise self._err()
        return buf[:min(n, got)]

    def tile_retile(self, rect):
        if self._lib.cs_tile_retile(self._engine, *[int(v) for v in rect]) != 0:
            raise self._err()
        self._agents_cache = None

    def tile_import(self, records):
        rec = np.ascontiguousarray(records, dtype=np.uint8)
        if len(rec) and self._lib.cs_tile_import(self._engine, rec.ctypes.data_as(C.c_void_p), len(rec)) != 0:
            raise self._err()
        self._agents_cache = None

    # -- tiles: route followers' set_target calls that missed the route book (cs_route_misses / _resolve) --
    def route_misses(self):
        n = self._lib.cs_route_misses(self._engine, None, 0)
        if n == 0:
            return []
        buf = (_abi.RouteMiss * n)()
        self._lib.cs_route_misses(self._engine, buf, n)
        return [(int(m.id), int(m.hlp), int(m.slot), m.px, m.py, m.tx, m.ty) for m in buf]

    def route_resolve(self, misses):
        buf = (_abi.RouteMiss * max(len(misses), 1))()
        for k, m in enumerate(misses):
            buf[k] = _abi.RouteMiss(*m)
        if self._lib.cs_route_resolve(self._engine, buf, len(misses)) != 0:
            raise self._err()

    # -- tiles: the RCCL transport of the C ABI (cs_rccl_*, cs_halo_exchange_rccl) --
    def rccl_unique_id(self):
        """Rank 0: the 128 bytes every rank passes to rccl_comm_init (ncclGetUniqueId)."""
        buf = (C.c_uint8 * _abi.CS_RCCL_UNIQUE_ID_BYTES)()
        if self._lib.cs_rccl_unique_id(buf) != 0:
            raise CrowdSimError("RCCL is not available (librccl.so.1)")
        return bytes(buf)

    def rccl_comm_init(self, n_ranks, rank, unique_id):
        buf = (C.c_uint8 * _abi.CS_RCCL_UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        if self._lib.cs_rccl_comm_init(self._engine, int(n_ranks), int(rank), buf) != 0:
            raise self._err()

    def halo_set_peers(self, peers):
        arr = (C.c_int32 * 8)(*[int(p) for p in peers])
        if self._lib.cs_halo_set_peers(self._engine, arr) != 0:
            raise self._err()

    def halo_exchange_rccl(self, axis=-1):
        if self._lib.cs_halo_exchange_rccl(self._engine, int(axis)) != 0:
            raise self._err()

    def allreduce_max_rccl(self, dev_ptr, n):
        if self._lib.cs_allreduce_max_i32_rccl(self._engine, C.c_void_p(dev_ptr), int(n)) != 0:
            raise self._err()

    def tile_step_rccl(self, dur, report=False):
        """One multi-GPU step of this tile in ONE call into the engine (cs_tile_step_rccl)."""
        dt = dur.total_seconds() if isinstance(dur, datetime.timedelta) else float(dur)
        rep = _abi.StepReport()
        rc = self._lib.cs_tile_step_rccl(self._engine, dt, C.byref(rep) if report else None)
        self._agents_cache = None
        if report:
            self.last_report = rep.as_dict()
        if rc != 0:
            raise self._err()

    def spawn_probe(self, dur):
        """Tile engines: which of MY source-sinks would spawn this step (uint8 flag per sink)."""
        dt = dur.total_seconds() if isinstance(dur, datetime.timedelta) else float(dur)
        # one flag per sink SLOT: a removed sink keeps its slot in the engine (registry.rs:16-21)
        flags = np.zeros(max(int(self._lib.cs_source_sink_slots(self._engine)), 1), dtype=np.uint8)
        n = self._lib.cs_spawn_probe(self._engine, dt, flags.ctypes.data_as(C.POINTER(C.c_uint8)),
                                     len(flags))
        if n == C.c_size_t(-1).value:
            raise self._err()
        return flags[:n]

    @property
    def device_bytes(self):
        """Device memory held by the engine (grows with capacity, never with steps or ids)."""
        return int(self._lib.cs_device_bytes(self._engine))

    def kernel_stat(self, which):
        """Diagnostics of the tiled kernel's work decomposition since creation (cs_kernel_stat):
        0 = windows that left the LDS path, 1 = windows walked in chunks; 2 / 3 = halo exchanges a tile with
        CS_CFG_TILE_OVERLAP issued ahead / could use."""
        return int(self._lib.cs_kernel_stat(self._engine, int(which)))

    @property
    def source_sink_slots(self):
        """Source-sink handles handed out so far (removed sinks keep their slot)."""
        return int(self._lib.cs_source_sink_slots(self._engine))

    def remove_agent_here(self, agent):
        """Tiles: remove `agent` if THIS engine holds it.  True = removed, False = not here;
        any other failure of the engine (poisoned, HIP error) raises."""
        rc = self._lib.cs_remove_agent(self._engine, int(agent))
        if rc == 0:
            self._agents_cache = None
            self._dispatch_events()
            return True
        if rc == 2:
            return False
        raise self._err()

    def spawn_commit(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        if self._lib.cs_spawn_commit(self._engine, flags.ctypes.data_as(C.POINTER(C.c_uint8)),
                                     len(flags)) != 0:
            raise self._err()

    # -- measurement --
    def profile_enable(self, kernel_mask=0xFFFFFFFF):
        """Bit k times kernel CS_K_k with hipEvents on the engine's stream; 0 = off."""
        self._lib.cs_profile_enable(self._engine, int(kernel_mask) & 0xFFFFFFFF)

    def profile_stride(self, every):
        """Time only every `every`-th launch of the enabled kernels."""
        self._lib.cs_profile_stride(self._engine, int(every))

    def profile_reset(self):
        self._lib.cs_profile_reset(self._engine)

    def profile_read(self):
        out = {}
        for k, name in enumerate(_abi.KERNEL_NAMES):
            ms, cnt = C.c_double(0), C.c_uint64(0)
            self._lib.cs_profile_read(self._engine, k, C.byref(ms), C.byref(cnt))
            out[name] = {"total_ms": ms.value, "launches": int(cnt.value)}
        return out
