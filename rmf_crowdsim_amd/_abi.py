"""ctypes declarations of include/crowdstep.h (the C-ABI boundary).

Declarations only: structs, callback types and `bind(lib)` which attaches
argument/return types to a loaded shared library.  No compute lives here.
"""
import ctypes as C

CS_ABI_VERSION = 1

CS_CFG_DEFAULT = 0
CS_CFG_FORCE_GATHER = 1
CS_CFG_FORCE_TILED = 2
CS_CFG_DENSE = 4
CS_CFG_TILE_OVERLAP = 8
CS_STAT_WINDOWS_OFF_LDS = 0
CS_STAT_WINDOWS_CHUNKED = 1
CS_STAT_EXCHANGES_AHEAD = 2
CS_STAT_EXCHANGES_AHEAD_USED = 3
CS_STAT_STEPS_ON_KEPT_WINDOWS = 4

CS_HLP_NONE, CS_HLP_CONSTANT, CS_HLP_ID_PARITY, CS_HLP_CALLBACK, CS_HLP_ROUTE = 0, 1, 2, 3, 4
CS_ROUTE_MAX_WAYPOINTS = 1023
CS_GEN_MONOTONIC, CS_GEN_POISSON_SEEDED, CS_GEN_CALLBACK = 0, 1, 2
CS_EVENT_SPAWNED, CS_EVENT_DESTROYED = 1, 2
(CS_K_NEIGHBOUR_FORCE, CS_K_SCAN, CS_K_SCATTER, CS_K_SPAWN, CS_K_HALO, CS_K_HALO_PACK, CS_K_HALO_EXCHANGE,
 CS_K_HALO_UNPACK, CS_K_STEP_BORDER, CS_K_STEP_INTERIOR, CS_K_COUNT) = range(11)
KERNEL_NAMES = ["neighbour_force", "scan", "scatter", "spawn", "halo", "halo_pack", "halo_exchange", "halo_unpack",
                "step_border", "step_interior"]
CS_DIR_XLO, CS_DIR_XHI, CS_DIR_YLO, CS_DIR_YHI = 0, 1, 2, 3
CS_DIR_XLO_YLO, CS_DIR_XLO_YHI, CS_DIR_XHI_YLO, CS_DIR_XHI_YHI = 4, 5, 6, 7
CS_HALO_RECORD_BYTES = 40
CS_RCCL_UNIQUE_ID_BYTES = 128
NO_SOURCE_SINK = 0xFFFFFFFF


class GridDesc(C.Structure):
    _fields_ = [("width", C.c_double), ("height", C.c_double), ("cell_size", C.c_double),
                ("offset_x", C.c_double), ("offset_y", C.c_double)]


class DeviceCfg(C.Structure):
    _fields_ = [("device_ordinal", C.c_int32), ("flags", C.c_uint32),
                ("tile_cx0", C.c_uint32), ("tile_cx1", C.c_uint32),
                ("tile_cy0", C.c_uint32), ("tile_cy1", C.c_uint32),
                ("halo_cells", C.c_uint32), ("reserved", C.c_uint32),
                ("capacity_hint", C.c_uint64), ("stream", C.c_void_p)]


class ZanlungoParams(C.Structure):
    _fields_ = [("agent_scale", C.c_double), ("obstacle_scale", C.c_double),
                ("reaction_time", C.c_double), ("force_distance", C.c_double),
                ("agent_mass", C.c_double), ("agent_radius", C.c_double)]


HlpVelocityFn = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64),
                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                            C.POINTER(C.c_double), C.POINTER(C.c_uint8))
HlpSetTargetFn = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_double,
                             C.c_double, C.c_double, C.c_double)
HlpRemoveFn = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64)
GeneratorFn = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_double)


class LpAgent(C.Structure):
    _fields_ = [("agent_id", C.c_uint64), ("x", C.c_double), ("y", C.c_double), ("vx", C.c_double),
                ("vy", C.c_double), ("preferred_vx", C.c_double), ("preferred_vy", C.c_double),
                ("eyesight_range", C.c_double), ("next_waypoint", C.c_uint64)]


LpBatchFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(LpAgent), C.POINTER(C.c_double),
                        C.POINTER(C.c_uint64), C.POINTER(LpAgent), C.POINTER(C.c_double))


RoutePlanFn = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                          C.POINTER(C.c_double), C.c_size_t)


class HlpDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("vx", C.c_double), ("vy", C.c_double),
                ("velocity", HlpVelocityFn), ("set_target", HlpSetTargetFn),
                ("remove_agent", HlpRemoveFn), ("user", C.c_void_p),
                ("route_plan", RoutePlanFn), ("route_scale", C.c_double),
                ("route_arrive", C.c_double), ("route_speed", C.c_double)]


class SnapshotRecord(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("vx", C.c_float), ("vy", C.c_float),
                ("id", C.c_uint32), ("next_waypoint", C.c_uint32)]


SNAPSHOT_DTYPE = [("x", "<f8"), ("y", "<f8"), ("vx", "<f4"), ("vy", "<f4"), ("id", "<u4"),
                  ("next_waypoint", "<u4")]


class SourceSinkDesc(C.Structure):
    _fields_ = [("source_x", C.c_double), ("source_y", C.c_double), ("radius_sink", C.c_double),
                ("generator_kind", C.c_uint32), ("rate", C.c_double), ("seed", C.c_uint64),
                ("generator", GeneratorFn), ("generator_user", C.c_void_p),
                ("hlp", C.c_uint32), ("lp", C.c_uint32),
                ("waypoints_xy", C.POINTER(C.c_double)), ("n_waypoints", C.c_size_t),
                ("loop_forever", C.c_int32), ("agent_eyesight_range", C.c_double)]


# cs_mesh_host_transport: a transport the host brings for a distributed mesh (host memory throughout)
MeshExchangeFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                             C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
MeshAllreduceMaxFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_size_t)
MeshAllgatherFn = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class MeshHostTransport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("exchange", MeshExchangeFn), ("allreduce_max_i32", MeshAllreduceMaxFn),
                ("allgather", MeshAllgatherFn)]


class MeshDesc(C.Structure):
    _fields_ = [("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32), ("halo_cells", C.c_uint32), ("flags", C.c_uint32),
                ("device_ordinal", C.c_int32), ("rank", C.c_int32), ("n_ranks", C.c_int32),
                ("density_per_cell", C.c_double), ("capacity_hint", C.c_uint64),
                ("weights_xy", C.POINTER(C.c_double)), ("n_weights", C.c_size_t),
                ("rccl_unique_id", C.POINTER(C.c_uint8)), ("host_transport", C.POINTER(MeshHostTransport))]


class RouteMiss(C.Structure):
    _fields_ = [("id", C.c_uint64), ("hlp", C.c_uint32), ("slot", C.c_uint32), ("px", C.c_double),
                ("py", C.c_double), ("tx", C.c_double), ("ty", C.c_double)]


class StepReport(C.Structure):
    _fields_ = [("n_agents", C.c_uint64), ("n_spawned", C.c_uint64), ("n_destroyed", C.c_uint64),
                ("n_waypoint_hits", C.c_uint64), ("n_tti_zero", C.c_uint64),
                ("n_nonfinite", C.c_uint64), ("n_clamped", C.c_uint64)]

    def as_dict(self):
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


class Event(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("source_sink", C.c_uint32), ("id", C.c_uint64),
                ("x", C.c_double), ("y", C.c_double)]


class AgentView(C.Structure):
    _fields_ = [("id", C.c_uint64), ("x", C.c_double), ("y", C.c_double),
                ("vx", C.c_double), ("vy", C.c_double), ("next_waypoint", C.c_uint64),
                ("eyesight_range", C.c_double)]


# name -> (restype, argtypes); every symbol include/crowdstep.h declares.
SYMBOLS = {
    "cs_abi_version": (C.c_uint32, []),
    "cs_create": (C.c_void_p, [C.POINTER(GridDesc), C.POINTER(DeviceCfg)]),
    "cs_destroy": (None, [C.c_void_p]),
    "cs_last_error": (C.c_char_p, [C.c_void_p]),
    "cs_backend_name": (C.c_char_p, [C.c_void_p]),
    "cs_register_zanlungo": (C.c_uint32, [C.c_void_p, C.POINTER(ZanlungoParams)]),
    "cs_register_no_local_plan": (C.c_uint32, [C.c_void_p]),
    "cs_register_lp_callback": (C.c_uint32, [C.c_void_p, LpBatchFn, C.c_void_p]),
    "cs_register_hlp": (C.c_uint32, [C.c_void_p, C.POINTER(HlpDesc)]),
    "cs_add_agents": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_uint32,
                                C.c_uint32, C.c_double, C.POINTER(C.c_uint64)]),
    "cs_remove_agent": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cs_add_source_sink": (C.c_uint32, [C.c_void_p, C.POINTER(SourceSinkDesc)]),
    "cs_remove_source_sink": (None, [C.c_void_p, C.c_uint32]),
    "cs_source_sink_slots": (C.c_size_t, [C.c_void_p]),
    "cs_device_bytes": (C.c_uint64, [C.c_void_p]),
    "cs_kernel_stat": (C.c_uint64, [C.c_void_p, C.c_uint32]),
    "cs_step": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(StepReport)]),
    "cs_synchronize": (C.c_int, [C.c_void_p]),
    "cs_agent_count": (C.c_size_t, [C.c_void_p]),
    "cs_read_agents": (C.c_size_t, [C.c_void_p, C.POINTER(AgentView), C.c_size_t]),
    "cs_snapshot_request": (C.c_int, [C.c_void_p]),
    "cs_snapshot_acquire": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.POINTER(SnapshotRecord)),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "cs_drain_events": (C.c_size_t, [C.c_void_p, C.POINTER(Event), C.c_size_t]),
    "cs_event_recording": (None, [C.c_void_p, C.c_int]),
    "cs_query_radius": (C.c_size_t, [C.c_void_p, C.c_double, C.c_double, C.c_double,
                                     C.POINTER(C.c_uint64), C.c_size_t]),
    "cs_query_knn": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_double, C.c_double,
                                  C.POINTER(C.c_uint64)]),
    "cs_query_radius_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "cs_query_knn_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.c_size_t,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]),
    "cs_profile_enable": (None, [C.c_void_p, C.c_uint32]),
    "cs_profile_stride": (None, [C.c_void_p, C.c_uint32]),
    "cs_profile_read": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double),
                                  C.POINTER(C.c_uint64)]),
    "cs_profile_reset": (None, [C.c_void_p]),
    "cs_halo_set_buffers": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_uint64]),
    "cs_halo_pack": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cs_halo_unpack": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cs_halo_pack_all": (C.c_int, [C.c_void_p]),
    "cs_halo_unpack_all": (C.c_int, [C.c_void_p]),
    "cs_spawn_probe": (C.c_size_t, [C.c_void_p, C.c_double, C.POINTER(C.c_uint8), C.c_size_t]),
    "cs_spawn_commit": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t]),
    "cs_spawn_probe_dev": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p, C.c_size_t]),
    "cs_spawn_commit_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cs_tile_histogram": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cs_tile_export": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cs_tile_retile": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "cs_tile_import": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cs_route_misses": (C.c_size_t, [C.c_void_p, C.POINTER(RouteMiss), C.c_size_t]),
    "cs_route_resolve": (C.c_int, [C.c_void_p, C.POINTER(RouteMiss), C.c_size_t]),
    "cs_rccl_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "cs_rccl_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "cs_rccl_comm_adopt": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cs_halo_set_peers": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "cs_halo_exchange_rccl": (C.c_int, [C.c_void_p, C.c_int32]),
    "cs_allreduce_max_i32_rccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cs_allgather_bytes_rccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cs_tile_step_rccl": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(StepReport)]),
    "cs_mesh_create": (C.c_void_p, [C.POINTER(GridDesc), C.POINTER(MeshDesc)]),
    "cs_mesh_destroy": (None, [C.c_void_p]),
    "cs_mesh_last_error": (C.c_char_p, [C.c_void_p]),
    "cs_mesh_local_tiles": (C.c_size_t, [C.c_void_p]),
    "cs_mesh_tile": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "cs_mesh_tile_rect": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "cs_mesh_register_zanlungo": (C.c_uint32, [C.c_void_p, C.POINTER(ZanlungoParams)]),
    "cs_mesh_register_no_local_plan": (C.c_uint32, [C.c_void_p]),
    "cs_mesh_register_hlp": (C.c_uint32, [C.c_void_p, C.POINTER(HlpDesc)]),
    "cs_mesh_register_lp_callback": (C.c_uint32, [C.c_void_p, LpBatchFn, C.c_void_p]),
    "cs_mesh_add_agents": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_uint32, C.c_uint32,
                                      C.c_double, C.POINTER(C.c_uint64)]),
    "cs_mesh_add_source_sink": (C.c_uint32, [C.c_void_p, C.POINTER(SourceSinkDesc)]),
    "cs_mesh_remove_source_sink": (None, [C.c_void_p, C.c_uint32]),
    "cs_mesh_remove_agent": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cs_mesh_event_recording": (None, [C.c_void_p, C.c_int]),
    "cs_mesh_drain_events": (C.c_size_t, [C.c_void_p, C.POINTER(Event), C.c_size_t]),
    "cs_mesh_step": (C.c_int, [C.c_void_p, C.c_double, C.POINTER(StepReport)]),
    "cs_mesh_synchronize": (C.c_int, [C.c_void_p]),
    "cs_mesh_agent_count": (C.c_size_t, [C.c_void_p]),
    "cs_mesh_read_agents": (C.c_size_t, [C.c_void_p, C.POINTER(AgentView), C.c_size_t]),
    "cs_mesh_tile_counts": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "cs_mesh_exchange_bytes": (C.c_uint64, [C.c_void_p]),
    "cs_mesh_recut": (C.c_int, [C.c_void_p]),
    "cs_mesh_query_radius_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cs_mesh_query_knn_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.c_size_t,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
}


def bind(lib):
    """Attach restype/argtypes for every ABI symbol; raises AttributeError if one is missing."""
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    version = lib.cs_abi_version()
    if version != CS_ABI_VERSION:
        raise RuntimeError(f"crowdstep ABI mismatch: library {version}, bindings {CS_ABI_VERSION}")
    return lib
