// NEVER COMPILED (no Rust toolchain in the build image).  CHECKED MECHANICALLY:
// tools/check_ffi_layout.py parses this file and include/crowdstep.h and asserts the same
// symbols, argument order and types, struct field order and types, constants, and (through a
// generated static_assert translation unit compiled with g++) the sizes and field offsets that
// #[repr(C)] gives these structs on x86-64.  Keep one item per line group as below: the checker
// reads this file with a small parser, not with rustc.
//
// Every item names the reference item it replaces in include/crowdstep.h.

#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const CS_ABI_VERSION: u32 = 1;

pub const CS_CFG_DEFAULT: u32 = 0;
pub const CS_CFG_FORCE_GATHER: u32 = 1;
pub const CS_CFG_FORCE_TILED: u32 = 2;
pub const CS_CFG_DENSE: u32 = 4;
pub const CS_CFG_TILE_OVERLAP: u32 = 8;
pub const CS_STAT_WINDOWS_OFF_LDS: u32 = 0;
pub const CS_STAT_WINDOWS_CHUNKED: u32 = 1;
pub const CS_STAT_EXCHANGES_AHEAD: u32 = 2;
pub const CS_STAT_EXCHANGES_AHEAD_USED: u32 = 3;
pub const CS_STAT_STEPS_ON_KEPT_WINDOWS: u32 = 4;

pub const CS_HLP_NONE: u32 = 0;
pub const CS_HLP_CONSTANT: u32 = 1;
pub const CS_HLP_ID_PARITY: u32 = 2;
pub const CS_HLP_CALLBACK: u32 = 3;
pub const CS_HLP_ROUTE: u32 = 4;
pub const CS_ROUTE_MAX_WAYPOINTS: usize = 1023;

pub const CS_GEN_MONOTONIC: u32 = 0;
pub const CS_GEN_POISSON_SEEDED: u32 = 1;
pub const CS_GEN_CALLBACK: u32 = 2;

pub const CS_EVENT_SPAWNED: u32 = 1;
pub const CS_EVENT_DESTROYED: u32 = 2;

pub const CS_K_NEIGHBOUR_FORCE: u32 = 0;
pub const CS_K_SCAN: u32 = 1;
pub const CS_K_SCATTER: u32 = 2;
pub const CS_K_SPAWN: u32 = 3;
pub const CS_K_HALO: u32 = 4;
pub const CS_K_HALO_PACK: u32 = 5;
pub const CS_K_HALO_EXCHANGE: u32 = 6;
pub const CS_K_HALO_UNPACK: u32 = 7;
pub const CS_K_STEP_BORDER: u32 = 8;
pub const CS_K_STEP_INTERIOR: u32 = 9;
pub const CS_K_COUNT: u32 = 10;

pub const CS_DIR_XLO: u32 = 0;
pub const CS_DIR_XHI: u32 = 1;
pub const CS_DIR_YLO: u32 = 2;
pub const CS_DIR_YHI: u32 = 3;
pub const CS_DIR_XLO_YLO: u32 = 4;
pub const CS_DIR_XLO_YHI: u32 = 5;
pub const CS_DIR_XHI_YLO: u32 = 6;
pub const CS_DIR_XHI_YHI: u32 = 7;
pub const CS_HALO_RECORD_BYTES: u32 = 40;
pub const CS_RCCL_UNIQUE_ID_BYTES: usize = 128;

/// Opaque engine handle (`struct cs_engine`).
#[repr(C)]
pub struct cs_engine {
    _private: [u8; 0],
}

/// Opaque handle of a crowd cut into spatial tiles (`struct cs_mesh`).
#[repr(C)]
pub struct cs_mesh {
    _private: [u8; 0],
}

/// A transport the host brings for a distributed mesh (MPI, sockets, ...): host memory throughout, collective calls.
pub type cs_mesh_exchange_fn = Option<unsafe extern "C" fn(user: *mut c_void, n: usize, peers: *const i32, send_tags: *const i32, recv_tags: *const i32, send_host: *const *const c_void, recv_host: *const *mut c_void, bytes: *const usize) -> c_int>;
pub type cs_mesh_allreduce_max_fn = Option<unsafe extern "C" fn(user: *mut c_void, values: *mut i32, n: usize) -> c_int>;
pub type cs_mesh_allgather_fn = Option<unsafe extern "C" fn(user: *mut c_void, mine: *const c_void, bytes: usize, all: *mut c_void) -> c_int>;
#[repr(C)]
#[derive(Clone, Copy)]
pub struct cs_mesh_host_transport {
    pub user: *mut c_void,
    pub exchange: cs_mesh_exchange_fn,
    pub allreduce_max_i32: cs_mesh_allreduce_max_fn,
    pub allgather: cs_mesh_allgather_fn,
}

/// How to cut and place a mesh (cs_mesh_create)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct cs_mesh_desc {
    pub tiles_x: u32,
    pub tiles_y: u32,
    pub halo_cells: u32,
    pub flags: u32,
    pub device_ordinal: i32,
    pub rank: i32,
    pub n_ranks: i32,
    pub density_per_cell: f64,
    pub capacity_hint: u64,
    pub weights_xy: *const f64,
    pub n_weights: usize,
    pub rccl_unique_id: *const u8,
    pub host_transport: *const cs_mesh_host_transport,
}

/// LocationHash2D::new(width, height, cell_size, offset), location_hash_2d.rs:33-51
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cs_grid_desc {
    pub width: f64,
    pub height: f64,
    pub cell_size: f64,
    pub offset_x: f64,
    pub offset_y: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cs_device_cfg {
    pub device_ordinal: i32,
    pub flags: u32,
    pub tile_cx0: u32,
    pub tile_cx1: u32,
    pub tile_cy0: u32,
    pub tile_cy1: u32,
    pub halo_cells: u32,
    pub reserved: u32,
    pub capacity_hint: u64,
    pub stream: *mut c_void,
}

/// Zanlungo::new(..), zanlungo.rs:31-48
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cs_zanlungo_params {
    pub agent_scale: f64,
    pub obstacle_scale: f64,
    pub reaction_time: f64,
    pub force_distance: f64,
    pub agent_mass: f64,
    pub agent_radius: f64,
}

pub type cs_hlp_velocity_fn = Option<unsafe extern "C" fn(user: *mut c_void, n: usize, ids: *const u64, pos_xy: *const f64, vel_xy: *const f64, time_s: f64, out_vel_xy: *mut f64, out_some: *mut u8)>;
pub type cs_hlp_set_target_fn = Option<unsafe extern "C" fn(user: *mut c_void, id: u64, pos_x: f64, pos_y: f64, point_x: f64, point_y: f64, tol_x: f64, tol_y: f64)>;
pub type cs_hlp_remove_fn = Option<unsafe extern "C" fn(user: *mut c_void, id: u64)>;
pub type cs_route_plan_fn = Option<unsafe extern "C" fn(user: *mut c_void, start_x: f64, start_y: f64, goal_x: f64, goal_y: f64, out_xy: *mut f64, cap: usize) -> usize>;
pub type cs_generator_fn = Option<unsafe extern "C" fn(user: *mut c_void, dt_seconds: f64) -> usize>;

/// The reference's Agent (lib.rs:47-65) as a host LocalPlanner sees it (slow path, cs_register_lp_callback)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct cs_lp_agent {
    pub agent_id: u64,
    pub x: f64,
    pub y: f64,
    pub vx: f64,
    pub vy: f64,
    pub preferred_vx: f64,
    pub preferred_vy: f64,
    pub eyesight_range: f64,
    pub next_waypoint: u64,
}
pub type cs_lp_batch_fn = Option<unsafe extern "C" fn(user: *mut c_void, n_agents: usize, agents: *const cs_lp_agent, recommended_xy: *const f64, nb_begin: *const u64, neighbours: *const cs_lp_agent, out_velocity_xy: *mut f64) -> c_int>;

/// HighLevelPlanner as data, highlevel_planners.rs:8-16
#[repr(C)]
#[derive(Clone, Copy)]
pub struct cs_hlp_desc {
    pub kind: u32,
    pub vx: f64,
    pub vy: f64,
    pub velocity: cs_hlp_velocity_fn,
    pub set_target: cs_hlp_set_target_fn,
    pub remove_agent: cs_hlp_remove_fn,
    pub user: *mut c_void,
    pub route_plan: cs_route_plan_fn,
    pub route_scale: f64,
    pub route_arrive: f64,
    pub route_speed: f64,
}

/// struct SourceSink, source_sink.rs:36-60
#[repr(C)]
#[derive(Clone, Copy)]
pub struct cs_source_sink_desc {
    pub source_x: f64,
    pub source_y: f64,
    pub radius_sink: f64,
    pub generator_kind: u32,
    pub rate: f64,
    pub seed: u64,
    pub generator: cs_generator_fn,
    pub generator_user: *mut c_void,
    pub hlp: u32,
    pub lp: u32,
    pub waypoints_xy: *const f64,
    pub n_waypoints: usize,
    pub loop_forever: i32,
    pub agent_eyesight_range: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct cs_step_report {
    pub n_agents: u64,
    pub n_spawned: u64,
    pub n_destroyed: u64,
    pub n_waypoint_hits: u64,
    pub n_tti_zero: u64,
    pub n_nonfinite: u64,
    pub n_clamped: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cs_event {
    pub kind: u32,
    pub source_sink: u32,
    pub id: u64,
    pub x: f64,
    pub y: f64,
}

/// pub struct Agent, lib.rs:46-65
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cs_agent_view {
    pub id: u64,
    pub x: f64,
    pub y: f64,
    pub vx: f64,
    pub vy: f64,
    pub next_waypoint: u64,
    pub eyesight_range: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cs_snapshot_record {
    pub x: f64,
    pub y: f64,
    pub vx: f32,
    pub vy: f32,
    pub id: u32,
    pub next_waypoint: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct cs_route_miss {
    pub id: u64,
    pub hlp: u32,
    pub slot: u32,
    pub px: f64,
    pub py: f64,
    pub tx: f64,
    pub ty: f64,
}

extern "C" {
    pub fn cs_abi_version() -> u32;
    pub fn cs_create(grid: *const cs_grid_desc, cfg: *const cs_device_cfg) -> *mut cs_engine;
    pub fn cs_destroy(e: *mut cs_engine);
    pub fn cs_last_error(e: *const cs_engine) -> *const c_char;
    pub fn cs_backend_name(e: *const cs_engine) -> *const c_char;
    pub fn cs_register_zanlungo(e: *mut cs_engine, p: *const cs_zanlungo_params) -> u32;
    pub fn cs_register_no_local_plan(e: *mut cs_engine) -> u32;
    pub fn cs_register_lp_callback(e: *mut cs_engine, f: cs_lp_batch_fn, user: *mut c_void) -> u32;
    pub fn cs_register_hlp(e: *mut cs_engine, d: *const cs_hlp_desc) -> u32;
    pub fn cs_add_agents(e: *mut cs_engine, xy: *const f64, n: usize, hlp: u32, lp: u32, eyesight: f64, out_ids: *mut u64) -> c_int;
    pub fn cs_remove_agent(e: *mut cs_engine, id: u64) -> c_int;
    pub fn cs_add_source_sink(e: *mut cs_engine, d: *const cs_source_sink_desc) -> u32;
    pub fn cs_remove_source_sink(e: *mut cs_engine, handle: u32);
    pub fn cs_source_sink_slots(e: *mut cs_engine) -> usize;
    pub fn cs_device_bytes(e: *mut cs_engine) -> u64;
    pub fn cs_kernel_stat(e: *mut cs_engine, which: u32) -> u64;
    pub fn cs_step(e: *mut cs_engine, dt_seconds: f64, report: *mut cs_step_report) -> c_int;
    pub fn cs_synchronize(e: *mut cs_engine) -> c_int;
    pub fn cs_agent_count(e: *mut cs_engine) -> usize;
    pub fn cs_read_agents(e: *mut cs_engine, out: *mut cs_agent_view, cap: usize) -> usize;
    pub fn cs_drain_events(e: *mut cs_engine, out: *mut cs_event, cap: usize) -> usize;
    pub fn cs_event_recording(e: *mut cs_engine, on: c_int);
    pub fn cs_snapshot_request(e: *mut cs_engine) -> c_int;
    pub fn cs_snapshot_acquire(e: *mut cs_engine, wait: c_int, out: *mut *const cs_snapshot_record, n: *mut usize, step_index: *mut u64) -> c_int;
    pub fn cs_query_radius(e: *mut cs_engine, radius: f64, x: f64, y: f64, out_ids: *mut u64, cap: usize) -> usize;
    pub fn cs_query_knn(e: *mut cs_engine, k: usize, x: f64, y: f64, out_ids: *mut u64) -> usize;
    pub fn cs_query_radius_batch(e: *mut cs_engine, n: usize, xy: *const f64, radius: *const f64, cap_per_query: usize, out_ids: *mut u64, out_counts: *mut u64, out_d2: *mut f32, out_cells: *mut u32) -> c_int;
    pub fn cs_query_knn_batch(e: *mut cs_engine, n: usize, xy: *const f64, k: usize, out_ids: *mut u64, out_counts: *mut u64, out_d2: *mut f32) -> c_int;
    pub fn cs_profile_enable(e: *mut cs_engine, kernel_mask: u32);
    pub fn cs_profile_stride(e: *mut cs_engine, every: u32);
    pub fn cs_profile_read(e: *mut cs_engine, kernel: u32, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn cs_profile_reset(e: *mut cs_engine);
    pub fn cs_halo_set_buffers(e: *mut cs_engine, dir: u32, send_dev: *mut c_void, recv_dev: *mut c_void, capacity_records: u64) -> c_int;
    pub fn cs_halo_pack(e: *mut cs_engine, axis: u32) -> c_int;
    pub fn cs_halo_unpack(e: *mut cs_engine, axis: u32) -> c_int;
    pub fn cs_halo_pack_all(e: *mut cs_engine) -> c_int;
    pub fn cs_halo_unpack_all(e: *mut cs_engine) -> c_int;
    pub fn cs_spawn_probe(e: *mut cs_engine, dt_seconds: f64, flags: *mut u8, cap: usize) -> usize;
    pub fn cs_spawn_commit(e: *mut cs_engine, flags: *const u8, n: usize) -> c_int;
    pub fn cs_spawn_probe_dev(e: *mut cs_engine, dt_seconds: f64, flags_dev: *mut c_int, cap: usize) -> c_int;
    pub fn cs_spawn_commit_dev(e: *mut cs_engine, flags_dev: *const c_int, n: usize) -> c_int;
    pub fn cs_tile_histogram(e: *mut cs_engine, rows: *mut u64, cols: *mut u64) -> c_int;
    pub fn cs_tile_export(e: *mut cs_engine, records: *mut c_void, cap_records: usize) -> usize;
    pub fn cs_tile_retile(e: *mut cs_engine, tile_cx0: u32, tile_cx1: u32, tile_cy0: u32, tile_cy1: u32) -> c_int;
    pub fn cs_tile_import(e: *mut cs_engine, records: *const c_void, n: usize) -> c_int;
    pub fn cs_route_misses(e: *mut cs_engine, out: *mut cs_route_miss, cap: usize) -> usize;
    pub fn cs_route_resolve(e: *mut cs_engine, all: *const cs_route_miss, n: usize) -> c_int;
    pub fn cs_rccl_unique_id(out_id: *mut u8) -> c_int;
    pub fn cs_rccl_comm_init(e: *mut cs_engine, n_ranks: i32, rank: i32, id: *const u8) -> c_int;
    pub fn cs_rccl_comm_adopt(e: *mut cs_engine, nccl_comm: *mut c_void) -> c_int;
    pub fn cs_halo_set_peers(e: *mut cs_engine, peers8: *const i32) -> c_int;
    pub fn cs_halo_exchange_rccl(e: *mut cs_engine, axis: i32) -> c_int;
    pub fn cs_allreduce_max_i32_rccl(e: *mut cs_engine, values_dev: *mut c_int, n: usize) -> c_int;
    pub fn cs_allgather_bytes_rccl(e: *mut cs_engine, send_dev: *const c_void, recv_dev: *mut c_void, bytes: usize) -> c_int;
    pub fn cs_tile_step_rccl(e: *mut cs_engine, dt_seconds: f64, report: *mut cs_step_report) -> c_int;

    pub fn cs_mesh_create(grid: *const cs_grid_desc, desc: *const cs_mesh_desc) -> *mut cs_mesh;
    pub fn cs_mesh_destroy(m: *mut cs_mesh);
    pub fn cs_mesh_last_error(m: *const cs_mesh) -> *const c_char;
    pub fn cs_mesh_local_tiles(m: *const cs_mesh) -> usize;
    pub fn cs_mesh_tile(m: *mut cs_mesh, local_index: usize) -> *mut cs_engine;
    pub fn cs_mesh_tile_rect(m: *const cs_mesh, local_index: usize, rect4: *mut u32) -> c_int;
    pub fn cs_mesh_register_zanlungo(m: *mut cs_mesh, p: *const cs_zanlungo_params) -> u32;
    pub fn cs_mesh_register_no_local_plan(m: *mut cs_mesh) -> u32;
    pub fn cs_mesh_register_hlp(m: *mut cs_mesh, d: *const cs_hlp_desc) -> u32;
    pub fn cs_mesh_register_lp_callback(m: *mut cs_mesh, f: cs_lp_batch_fn, user: *mut c_void) -> u32;
    pub fn cs_mesh_add_agents(m: *mut cs_mesh, xy: *const f64, n: usize, hlp: u32, lp: u32, eyesight: f64, out_ids: *mut u64) -> c_int;
    pub fn cs_mesh_add_source_sink(m: *mut cs_mesh, d: *const cs_source_sink_desc) -> u32;
    pub fn cs_mesh_remove_source_sink(m: *mut cs_mesh, handle: u32);
    pub fn cs_mesh_remove_agent(m: *mut cs_mesh, id: u64) -> c_int;
    pub fn cs_mesh_event_recording(m: *mut cs_mesh, on: c_int);
    pub fn cs_mesh_drain_events(m: *mut cs_mesh, out: *mut cs_event, cap: usize) -> usize;
    pub fn cs_mesh_step(m: *mut cs_mesh, dt_seconds: f64, report: *mut cs_step_report) -> c_int;
    pub fn cs_mesh_synchronize(m: *mut cs_mesh) -> c_int;
    pub fn cs_mesh_agent_count(m: *mut cs_mesh) -> usize;
    pub fn cs_mesh_read_agents(m: *mut cs_mesh, out: *mut cs_agent_view, cap: usize) -> usize;
    pub fn cs_mesh_tile_counts(m: *mut cs_mesh, out_per_local_tile: *mut u64) -> c_int;
    pub fn cs_mesh_exchange_bytes(m: *const cs_mesh) -> u64;
    pub fn cs_mesh_recut(m: *mut cs_mesh) -> c_int;
    pub fn cs_mesh_query_radius_batch(m: *mut cs_mesh, n: usize, xy: *const f64, radius: *const f64, cap_per_query: usize, out_ids: *mut u64, out_counts: *mut u64) -> c_int;
    pub fn cs_mesh_query_knn_batch(m: *mut cs_mesh, n: usize, xy: *const f64, k: usize, out_ids: *mut u64, out_counts: *mut u64) -> c_int;
}
