/*
 * crowdstep.h — C ABI of the MI355X-native crowd-step engine.
 *
 * This is the drop-in boundary for the `Simulation::step` hot path of
 * open-rmf/rmf_crowdsim.  The reference exposes a Rust trait/struct API and no
 * FFI of its own, so every entry point below names the reference item it
 * replaces (paths relative to the reference tree, rmf_crowdsim/src/...).
 * A Rust host keeps its `Simulation` / `HighLevelPlanner` / `LocalPlanner` /
 * `EventListener` surface and forwards to these symbols through `extern "C"`
 * (binding sketch: INTEGRATION.md).
 *
 * Two shared libraries implement this header:
 *   - rmf_crowdsim_amd/lib/libcrowdstep_hip.so  : the product (HIP kernels, gfx950)
 *   - oracle/_build/libcrowdstep_oracle.so      : TEST ORACLE ONLY (f64 CPU restatement)
 *
 * Conventions
 *   - plain pointers and sizes, no C++/torch types; the engine owns all device
 *     memory, the caller owns every buffer it passes in.
 *   - `int` results: 0 = Ok, non-zero = Err; `cs_last_error` then carries the
 *     reference's error string ("Index out of bounds",
 *     "Failed to add agents from source").
 *   - one engine = one caller thread at a time (mirrors `&mut self`, lib.rs:195).
 *   - positions/velocities cross the ABI as f64 (the reference's `Vec2f`,
 *     lib.rs:40-43); the device state is f32 cell-relative.
 *   - agent ids cross as u64 (`AgentId = usize`, lib.rs:36).
 */
#ifndef CROWDSTEP_H
#define CROWDSTEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_ABI_VERSION 1

typedef struct cs_engine cs_engine;

/* LocationHash2D::new(width, height, cell_size, offset)
 * spatial_index/location_hash_2d.rs:33-51 */
typedef struct cs_grid_desc {
  double width;
  double height;
  double cell_size;
  double offset_x;
  double offset_y;
} cs_grid_desc;

/* Device placement.  A tile engine owns the cells [tile_cx0, tile_cx1) x
 * [tile_cy0, tile_cy1) of the global grid (x = the index location_to_index
 * multiplies by the row stride, location_hash_2d.rs:59) and keeps a ghost ring
 * of halo_cells cells for neighbour queries (multi-GPU, DESIGN.md "Tiles");
 * halo_cells must be >= ceil(max eyesight / cell_size).
 * All-zero tile bounds = the engine owns the whole grid. */
typedef struct cs_device_cfg {
  int32_t device_ordinal;   /* HIP device index for this engine               */
  uint32_t flags;           /* CS_CFG_* bits                                  */
  uint32_t tile_cx0, tile_cx1, tile_cy0, tile_cy1;
  uint32_t halo_cells;
  uint32_t reserved;
  uint64_t capacity_hint;   /* expected max agents (0 = grow on demand)       */
  void* stream;             /* hipStream_t to run on (NULL = engine's own)    */
} cs_device_cfg;

#define CS_CFG_DEFAULT 0u
#define CS_CFG_FORCE_GATHER 1u  /* use the direct-gather neighbour kernel only */
#define CS_CFG_FORCE_TILED 2u   /* use the LDS-tiled neighbour kernel only     */
#define CS_CFG_DENSE 4u         /* expect more than 64 neighbours in sight somewhere (hotspots):
                                 * neighbour lists of up to 128 entries (more scratch memory);
                                 * chosen automatically when the MEAN occupancy says so */
#define CS_CFG_TILE_OVERLAP 8u  /* tile engines driven by cs_tile_step_rccl: the windows along the tile's
                                 * edges run as a launch of their own on a second stream and pack the next
                                 * step's halo records; that exchange then runs while the interior windows
                                 * are still being stepped (SURVEY.md section 8e "overlap with K4 on
                                 * interior cells").  Calls that change agents between two steps
                                 * (cs_add_agents, cs_remove_agent, cs_tile_import ...) must then be made on
                                 * every rank alike: they void the exchange made ahead, and all ranks have
                                 * to repeat it together.  An agent of an interior window that lands in the halo
                                 * band within one step fails the step (it would miss the exchange): one that
                                 * moves more than a cell per step, or whose position turns NaN (the reference
                                 * bins NaN to cell 0, location_hash_2d.rs:54-66, which on a tile is a ghost cell). */

/* Zanlungo::new(agent_scale, obstacle_scale, reaction_time, force_distance,
 *               agent_mass, agent_radius)   local_planners/zanlungo.rs:31-48 */
typedef struct cs_zanlungo_params {
  double agent_scale;
  double obstacle_scale; /* stored, never read by the reference */
  double reaction_time;  /* stored, never read by the reference */
  double force_distance;
  double agent_mass;
  double agent_radius;
} cs_zanlungo_params;

/* HighLevelPlanner (highlevel_planners/highlevel_planners.rs:8-16).
 * Trait objects cannot run on the device, so planners are registered as data:
 *   NONE      get_desired_velocity returns None      (lib.rs:263-273: vel = 0)
 *   CONSTANT  Some(v)            (StubHighLevelPlan, lib.rs:391-420)
 *   ID_PARITY even id -> Some(-v), odd id -> Some(v)
 *             (rmf_crowdsim_viz/src/main.rs:20-30)
 *   CALLBACK  host trait object, evaluated per step for the whole batch
 *   ROUTE     RMFPlanner's follower (rmf/mod.rs:195-242) as data: the route of
 *             an agent is planned on the host ONCE per set_target (A* itself is
 *             the host's business, `route_plan`), cached per (start, goal) hash
 *             pair like route_plans_by_location (rmf/mod.rs:86-88,217-236); per
 *             step the device returns unit(route[wp] - position) * route_speed
 *             and moves on to the next waypoint inside route_arrive (:203-209).
 *             Agents that never got a target return None (:211-214).          */
enum {
  CS_HLP_NONE = 0,
  CS_HLP_CONSTANT = 1,
  CS_HLP_ID_PARITY = 2,
  CS_HLP_CALLBACK = 3,
  CS_HLP_ROUTE = 4
};

/* get_desired_velocity(&mut self, &Agent, Duration) -> Option<Vec2f> for n
 * agents at once.  `time_s` is always 0 (the reference never advances
 * sim_time, lib.rs:81,110,268).  out_some[i]=0 means None. */
typedef void (*cs_hlp_velocity_fn)(void* user, size_t n, const uint64_t* ids,
                                   const double* pos_xy, const double* vel_xy,
                                   double time_s, double* out_vel_xy,
                                   uint8_t* out_some);
/* set_target(&mut self, &Agent, point, tolerance)  highlevel_planners.rs:12 */
typedef void (*cs_hlp_set_target_fn)(void* user, uint64_t id, double pos_x,
                                     double pos_y, double point_x,
                                     double point_y, double tol_x, double tol_y);
/* remove_agent_id(&mut self, AgentId)              highlevel_planners.rs:15 */
typedef void (*cs_hlp_remove_fn)(void* user, uint64_t id);
/* RMFPlanner::plan_route(start, goal) -> Option<Vec<Vec2f>>   rmf/mod.rs:160-192
 * writes up to `cap` waypoints (x, y pairs, the goal last) and returns their
 * number; 0 = "Failed to find contiguous path" (the agent then has no route). */
typedef size_t (*cs_route_plan_fn)(void* user, double start_x, double start_y,
                                   double goal_x, double goal_y, double* out_xy,
                                   size_t cap);
#define CS_ROUTE_MAX_WAYPOINTS 1023

typedef struct cs_hlp_desc {
  uint32_t kind; /* CS_HLP_* */
  double vx, vy; /* CONSTANT / ID_PARITY */
  cs_hlp_velocity_fn velocity; /* CALLBACK */
  cs_hlp_set_target_fn set_target;
  cs_hlp_remove_fn remove_agent;
  void* user;
  /* ROUTE */
  cs_route_plan_fn route_plan;
  double route_scale;  /* SpatialHash resolution of the route cache, rmf/mod.rs:70-77 */
  double route_arrive; /* 1e-1 in the reference, rmf/mod.rs:203                      */
  double route_speed;  /* 1.0 in the reference (a unit vector), rmf/mod.rs:209       */
} cs_hlp_desc;

/* CrowdGenerator::get_number_to_spawn(&self, Duration) -> usize
 * source_sink/source_sink.rs:30-33.
 *   MONOTONIC       round(dt * rate)                     source_sink.rs:96-100
 *   POISSON_SEEDED  Poisson(dt * rate) drawn from a counter-based generator
 *                   keyed by (seed, step index); replaces PoissonCrowd
 *                   (source_sink.rs:75-82), whose thread_rng is unseedable
 *   CALLBACK        host trait object                                        */
enum { CS_GEN_MONOTONIC = 0, CS_GEN_POISSON_SEEDED = 1, CS_GEN_CALLBACK = 2 };
typedef size_t (*cs_generator_fn)(void* user, double dt_seconds);

/* A LocalPlanner that is host code: trait LocalPlanner, local_planners/local_planner.rs:7-18, called at
 * lib.rs:276-291.  THE SLOW PATH: the two planners the reference ships run on the device
 * (cs_register_zanlungo / cs_register_no_local_plan); any other one is evaluated on the host, batched per planner
 * and step.  cs_lp_agent is the reference's Agent (lib.rs:47-65) as `step` maintains it (orientation and
 * angular_vel are never written after creation, lib.rs:138,141).  The callback gets, for every agent of the
 * planner that this engine steps, in ascending id:
 *   agents[k]               the agent as it was at the start of the step, preferred_vel = the velocity its
 *                           high-level planner recommends (the clone of lib.rs:261-271)
 *   recommended_xy[2k..]    that velocity again: the third argument of get_desired_velocity
 *   neighbours[nb_begin[k] .. nb_begin[k + 1])   the agents within agents[k].eyesight_range of it (strict <,
 *                           itself excluded: lib.rs:279-287), start-of-step state, preferred_vel = (0, 0) as in the
 *                           reference, in the canonical order of the step (cells
 *                           x-major / y-minor, ascending id inside a cell)
 * and writes the agent's new velocity to out_velocity_xy[2k..]. */
typedef struct cs_lp_agent {
  uint64_t agent_id;
  double x, y;
  double vx, vy;
  double preferred_vx, preferred_vy;
  double eyesight_range;
  uint64_t next_waypoint;
} cs_lp_agent;
/* Returns 0, or non-zero when the planner could not answer (an exception on the host's side, a poisoned lock): the step
 * then fails with Err("a host LocalPlanner failed") and commits nothing, instead of integrating velocities nobody gave. */
typedef int (*cs_lp_batch_fn)(void* user, size_t n_agents, const cs_lp_agent* agents, const double* recommended_xy,
                              const uint64_t* nb_begin, const cs_lp_agent* neighbours, double* out_velocity_xy);

/* struct SourceSink                              source_sink/source_sink.rs:36-60 */
typedef struct cs_source_sink_desc {
  double source_x, source_y;
  double radius_sink;
  uint32_t generator_kind; /* CS_GEN_* */
  double rate;
  uint64_t seed;
  cs_generator_fn generator;
  void* generator_user;
  uint32_t hlp; /* handle from cs_register_hlp */
  uint32_t lp;  /* handle from cs_register_zanlungo / _no_local_plan */
  const double* waypoints_xy; /* n_waypoints pairs; last one is the sink */
  size_t n_waypoints;
  int32_t loop_forever;
  double agent_eyesight_range;
} cs_source_sink_desc;

/* What one step did (the reference prints or drops these). */
typedef struct cs_step_report {
  uint64_t n_agents;        /* alive after the step                            */
  uint64_t n_spawned;       /* lib.rs:199-254                                  */
  uint64_t n_destroyed;     /* lib.rs:378-380                                  */
  uint64_t n_waypoint_hits; /* "Reached waypoint", lib.rs:317                  */
  uint64_t n_tti_zero;      /* agents whose min time-to-collision was 0        */
  uint64_t n_nonfinite;     /* agents whose new state is NaN/inf               */
  uint64_t n_clamped;       /* agents binned by the saturating cast (a4)       */
} cs_step_report;

/* EventListener callbacks, queued during cs_step and drained afterwards
 * lib.rs:22-33,151-153,189-191 */
enum { CS_EVENT_SPAWNED = 1, CS_EVENT_DESTROYED = 2 };
typedef struct cs_event {
  uint32_t kind;
  uint32_t source_sink; /* owner handle or UINT32_MAX */
  uint64_t id;
  double x, y; /* spawn position (SPAWNED only) */
} cs_event;

/* pub struct Agent, lib.rs:46-65 (fields a caller can observe) */
typedef struct cs_agent_view {
  uint64_t id;
  double x, y;
  double vx, vy;
  uint64_t next_waypoint;
  double eyesight_range;
} cs_agent_view;

/* ---- lifetime ---------------------------------------------------------- */
uint32_t cs_abi_version(void);
/* Simulation::new(LocationHash2D::new(..))   lib.rs:103, location_hash_2d.rs:33 */
cs_engine* cs_create(const cs_grid_desc* grid, const cs_device_cfg* cfg);
void cs_destroy(cs_engine*);
/* The message of the last Err of this engine; with a null engine, why the calling thread's
 * last cs_create returned null (no device, grid beyond 32-bit cell indices, allocation). */
const char* cs_last_error(const cs_engine*);
/* "hip:<gcnArchName>" for the product, "oracle:f64" for the test oracle */
const char* cs_backend_name(const cs_engine*);

/* ---- planners as data -------------------------------------------------- */
uint32_t cs_register_zanlungo(cs_engine*, const cs_zanlungo_params*); /* zanlungo.rs:31 */
uint32_t cs_register_no_local_plan(cs_engine*);           /* no_local_plan.rs:7-18 */
/* any other `impl LocalPlanner`: evaluated on the host (see cs_lp_batch_fn)   local_planner.rs:7-18 */
uint32_t cs_register_lp_callback(cs_engine*, cs_lp_batch_fn fn, void* user);
uint32_t cs_register_hlp(cs_engine*, const cs_hlp_desc*); /* highlevel_planners.rs:8 */

/* ---- population -------------------------------------------------------- */
/* Simulation::add_agents                                        lib.rs:119-156 */
int cs_add_agents(cs_engine*, const double* xy, size_t n, uint32_t hlp,
                  uint32_t lp, double eyesight, uint64_t* out_ids);
/* Simulation::remove_agents (unknown id: returns Err instead of panicking)
 *                                                               lib.rs:176-192 */
int cs_remove_agent(cs_engine*, uint64_t id);
/* Simulation::add_source_sink / remove_source_sink              lib.rs:159-168
 * Returns the handle, or UINT32_MAX (cs_last_error says why).  On a tile engine the first leg of a
 * sink whose planner is CS_HLP_ROUTE is planned here (route_plan is called once per tile, in
 * registration order) so that every tile numbers routes alike; see cs_route_resolve for later legs. */
uint32_t cs_add_source_sink(cs_engine*, const cs_source_sink_desc*);
void cs_remove_source_sink(cs_engine*, uint32_t handle);
/* Number of source-sink handles ever handed out (registry.rs:16-21: ids only grow; a removed
 * sink keeps its slot).  cs_spawn_probe / cs_spawn_commit take one flag per SLOT. */
size_t cs_source_sink_slots(cs_engine*);

/* ---- the hot path ------------------------------------------------------ */
/* Simulation::step(dur), dt_seconds = dur.as_secs_f64()          lib.rs:195-383
 * report may be NULL (then the call does not wait for the device unless
 * source-sinks or callback planners need the host).  The Err of a step that was not
 * waited for ("Index out of bounds") is returned by the next call that does wait:
 * cs_synchronize, a cs_step with a report (or one that needs the host), and it makes
 * cs_read_agents return 0 once (cs_last_error says why); later steps are refused. */
int cs_step(cs_engine*, double dt_seconds, cs_step_report* report);
/* Wait until every queued step has finished on the device; 0, or the Err of one of them. */
int cs_synchronize(cs_engine*);

/* ---- observation ------------------------------------------------------- */
size_t cs_agent_count(cs_engine*);                       /* agents.len(), lib.rs:71 */
/* `pub agents` view, ascending id; returns number written       lib.rs:71    */
size_t cs_read_agents(cs_engine*, cs_agent_view* out, size_t cap);
size_t cs_drain_events(cs_engine*, cs_event* out, size_t cap);
/* Events are queued only while recording is on (default on).  A host without listeners
 * (lib.rs:88 registry empty) turns it off so the queue cannot grow. */
void cs_event_recording(cs_engine*, int on);
/* Streaming `agents` view for a renderer / embedder (lib.rs:71 read every frame,
 * rmf_crowdsim_viz/src/main.rs:112-128).  cs_snapshot_request queues, behind the
 * steps already queued, a device-side gather of the live agents (global f64
 * positions) and its copy into pinned host memory on a second stream; it does
 * not wait, so the next cs_step overlaps the transfer.  Two buffers alternate.
 * cs_snapshot_acquire hands out the most recent requested snapshot:
 *   0 = out, n and step_index are filled; the memory stays valid until the second
 *       cs_snapshot_request from now
 *   1 = nothing was requested, 2 = not complete yet (only with wait == 0),
 *   3 = device error (cs_last_error)
 * Agents come in no particular order (cs_read_agents sorts by id).            */
typedef struct cs_snapshot_record {
  double x, y;
  float vx, vy;
  uint32_t id;
  uint32_t next_waypoint;
} cs_snapshot_record;
int cs_snapshot_request(cs_engine*);
int cs_snapshot_acquire(cs_engine*, int wait, const cs_snapshot_record** out,
                        size_t* n, uint64_t* step_index);
/* SpatialIndex::get_neighbours_in_radius                location_hash_2d.rs:240-258
 * returns the full count; writes min(count, cap) ids in reference cell order
 * (x-major, y-minor) with ascending id inside a cell.  As in the reference, a rectangle wider
 * than the grid lists the members of a cell once per row through which it reaches the cell
 * (y runs past the row stride, :74-85): an id can come back several times. */
size_t cs_query_radius(cs_engine*, double radius, double x, double y,
                       uint64_t* out_ids, size_t cap);
/* SpatialIndex::get_nearest_neighbours                  location_hash_2d.rs:151-238 */
size_t cs_query_knn(cs_engine*, size_t k, double x, double y, uint64_t* out_ids);

/* Batch forms: n queries in one launch (one wave per query).  out_ids holds cap_per_query entries
 * per query (query i at out_ids + i * cap_per_query, same order as cs_query_radius), out_counts[i]
 * the full count of query i; out_d2 (squared distances, f32) and out_cells (the GLOBAL cell of each
 * hit, x * (width / cell) + y) are optional (NULL).  On a TILE engine these calls are available and
 * report the agents the tile HOLDS (after a step: everything it stepped, also what has just walked
 * into its ring; after a halo exchange: what lies in its owned cells); every agent is held by exactly
 * one tile, so the union over the tiles of a mesh, ordered by (cell, id), is the reference's answer.
 * cs_query_knn_batch: out_ids / out_d2 hold k entries per query, out_counts[i] <= k. */
int cs_query_radius_batch(cs_engine*, size_t n, const double* xy, const double* radius, size_t cap_per_query,
                          uint64_t* out_ids, uint64_t* out_counts, float* out_d2, uint32_t* out_cells);
int cs_query_knn_batch(cs_engine*, size_t n, const double* xy, size_t k, uint64_t* out_ids, uint64_t* out_counts,
                       float* out_d2);

/* Device memory the engine holds, in bytes.  It grows with the agent CAPACITY (slots), the grid and
 * the number of source-sinks and routes, never with the number of steps or of ids handed out. */
uint64_t cs_device_bytes(cs_engine*);

/* How the LDS-tiled neighbour kernel has fared since the engine was created (waits for the device):
 * CS_STAT_WINDOWS_OFF_LDS  band windows whose staged tile did not fit the LDS (or the fixed-point
 *                          range) and whose agents therefore took the gather path, one slow workgroup
 *                          each: the window builder bounds a window by what it stages, so this stays 0
 *                          unless single cells are overfull;
 * CS_STAT_WINDOWS_CHUNKED  windows that held more than a workgroup's 256 agents and were walked in
 *                          chunks.
 * Diagnostics of the engine's own work decomposition; the reference has no counterpart. */
#define CS_STAT_WINDOWS_OFF_LDS 0u
#define CS_STAT_WINDOWS_CHUNKED 1u
/* halo exchanges a tile engine with CS_CFG_TILE_OVERLAP issued AHEAD (cs_tile_step_rccl: on its second stream, behind the
 * border windows' launch, for the next step) and how many of those the next step could use (the others were made void
 * by a change of the agents in between) */
#define CS_STAT_EXCHANGES_AHEAD 2u
#define CS_STAT_EXCHANGES_AHEAD_USED 3u
/* steps of a small crowd (at most 300,000 slots) that ran on the band windows cut ONE STEP EARLIER: the builder runs in
 * every step, as workgroups leading the neighbour kernel's own launch, and cuts the windows the NEXT step runs on (cut to
 * tile their band completely, with room to spare, so they stay correct while the agents move one step).  The environment
 * variable CS_WINDOWS_KEEP is a switch, not a period: 0 = cut every step's windows in its own scatter launch */
#define CS_STAT_STEPS_ON_KEPT_WINDOWS 4u
uint64_t cs_kernel_stat(cs_engine*, uint32_t which);

/* ---- measurement (bench.py / rocprof cross-check) ---------------------- */
/* Kernel names the engine launches per step, for HIP-event timing. */
enum {
  CS_K_NEIGHBOUR_FORCE = 0, /* Zanlungo neighbour pass + integrate (K4)  */
  CS_K_SCAN = 1,            /* exclusive scan of cell counts (K2)        */
  CS_K_SCATTER = 2,         /* reorder into cell order (K3)              */
  CS_K_SPAWN = 3,           /* source occupancy + append (K6)            */
  CS_K_HALO = 4,            /* halo pack/unpack of the two-phase schedule (K7: cs_halo_pack / cs_halo_unpack) */
  /* the phases of a tile's step in the one-phase schedule (cs_tile_step_rccl, cs_mesh_step), each timed by itself: */
  CS_K_HALO_PACK = 5,       /* cs_halo_pack_all's launch (none when the step kernel packed: the usual case)     */
  CS_K_HALO_EXCHANGE = 6,   /* cs_halo_exchange_rccl: the ncclSend / ncclRecv group, on the stream it was issued on
                             * (the second stream for an exchange made ahead under CS_CFG_TILE_OVERLAP)           */
  CS_K_HALO_UNPACK = 7,     /* cs_halo_unpack_all's launch                                                       */
  CS_K_STEP_BORDER = 8,     /* CS_CFG_TILE_OVERLAP: the neighbour kernel's launch over the windows along the
                             * tile's edges (on the second stream when the exchange follows it there) ...        */
  CS_K_STEP_INTERIOR = 9,   /* ... and over the interior windows (CS_K_NEIGHBOUR_FORCE spans both)               */
  CS_K_COUNT = 10
};
/* Per-kernel hipEvent timing: bit k of kernel_mask times kernel CS_K_k (events are
 * recorded on the engine's stream around each launch); 0 turns timing off. */
void cs_profile_enable(cs_engine*, uint32_t kernel_mask);
/* Time only every `every`-th launch of an enabled kernel (default 1 = all): an event pair costs
 * the stream a few microseconds per launch.  cs_profile_read then reports the timed launches. */
void cs_profile_stride(cs_engine*, uint32_t every);
/* Sum of durations (ms) and launch count since the last reset. */
int cs_profile_read(cs_engine*, uint32_t kernel, double* total_ms, uint64_t* launches);
void cs_profile_reset(cs_engine*);

/* ---- tiles: halo exchange hooks (multi-GPU, one engine per rank) -------- */
/* The reference has no counterpart (single process).  Per step and tile:
 *   cs_halo_pack_all -> move the eight send buffers to the neighbours' recv buffers
 *   -> cs_halo_unpack_all -> cs_step
 * or, in two phases with edge buffers only,
 *   cs_halo_pack(0) -> move XLO/XHI send buffers to the neighbours' recv buffers
 *   -> cs_halo_unpack(0) -> cs_halo_pack(1) -> move YLO/YHI -> cs_halo_unpack(1)
 *   -> cs_step.   The transport is the caller's (RCCL send/recv, torch.distributed,
 * hipMemcpyPeer): the engine only fills and drains device buffers, on its stream.
 * In tile mode cs_add_agents takes the GLOBAL position list on every tile (ids are
 * allocated for all of it, agents outside the owned cells are skipped). */
enum { CS_DIR_XLO = 0, CS_DIR_XHI = 1, CS_DIR_YLO = 2, CS_DIR_YHI = 3,
       /* the diagonal neighbours, for the one-phase exchange (cs_halo_pack_all) */
       CS_DIR_XLO_YLO = 4, CS_DIR_XLO_YHI = 5, CS_DIR_XHI_YLO = 6, CS_DIR_XHI_YHI = 7 };
/* A record: cell-relative offset and velocity (4 x f32), id, group | next_waypoint << 16, the
 * GLOBAL cell (x row, y column; 2 x u32), the route follower's state and a reserved word.  The
 * layout is the engine's business: the transport moves bytes. */
#define CS_HALO_RECORD_BYTES 40u
/* Caller-provided device buffers (e.g. torch CUDA tensors) of
 * (capacity_records + 1) * CS_HALO_RECORD_BYTES bytes: record 0 is the header
 * (word 0 = record count).  A direction without a neighbour tile gets no buffers. */
int cs_halo_set_buffers(cs_engine*, uint32_t dir, void* send_dev, void* recv_dev,
                        uint64_t capacity_records);
/* Fill the two send buffers of one axis (0 = X pair, 1 = Y pair) with every agent
 * within 2 * halo_cells of the shared edge. */
int cs_halo_pack(cs_engine*, uint32_t axis);
/* Merge what arrived on one axis into the tile (as owned agents or ghosts, by cell). */
int cs_halo_unpack(cs_engine*, uint32_t axis);
/* One-phase form of the same exchange: with buffers for all eight neighbours (edges and
 * corners) cs_halo_pack_all fills every send buffer in one launch (an agent near a corner goes
 * to the diagonal tile directly instead of being forwarded), one transport round moves them,
 * cs_halo_unpack_all merges the eight receive buffers.  Same result as the two-phase form. */
int cs_halo_pack_all(cs_engine*);
int cs_halo_unpack_all(cs_engine*);
/* Source-sinks on tiles: every tile registers ALL source-sinks (same order); agent ids must
 * follow the global sink order (lib.rs:199-254), so Phase A is split.  After the halo exchange:
 *   cs_spawn_probe   flags[s] = 1 iff this tile owns sink s, its generator fired and nobody
 *                    (owned agent or ghost) stands within 0.4 of its source; returns the
 *                    number of sinks (SIZE_MAX on error)
 *   (the caller ORs the flag vectors of all tiles: one small all-reduce)
 *   cs_spawn_commit  assigns ids in ascending handle order over the combined flags and appends
 *                    the agents of the sinks this tile owns
 * then cs_step.  Generators must be deterministic across tiles (MONOTONIC, POISSON_SEEDED). */
size_t cs_spawn_probe(cs_engine*, double dt_seconds, uint8_t* flags, size_t cap);
int cs_spawn_commit(cs_engine*, const uint8_t* flags, size_t n);
/* The same two calls with the flags in device memory (one int per sink) and no wait for the
 * host: the probe is a kernel, the caller all-reduces (max) the device vectors, the commit
 * kernel hands out the ids from the engine's device-side counter.  For hosts without listeners,
 * callback planners and per-step reports; the host-side pair above remains for those. */
int cs_spawn_probe_dev(cs_engine*, double dt_seconds, int* flags_dev, size_t cap);
int cs_spawn_commit_dev(cs_engine*, const int* flags_dev, size_t n);

/* Re-cutting a running mesh (a clustered crowd drifts: BASELINE.json configs[4]).  Between two
 * steps: cs_tile_histogram adds the agents the tile holds per global x-row (height / cell entries)
 * and y-column (width / cell entries) into the caller's arrays (summed over the tiles they give the
 * new cuts); cs_tile_export writes every agent the tile holds as a halo record (CS_HALO_RECORD_BYTES each,
 * host memory) and returns their number (SIZE_MAX on error; call with cap 0 to size the buffer);
 * cs_tile_retile gives the engine a new owned rectangle: it forgets its agents and ghosts and its
 * halo buffers (set them again: the edges moved), keeps planners, groups, source-sinks, the id
 * counter and the routes; cs_tile_import takes, from the exports of ALL tiles, the records whose
 * cell the engine now owns.  The next halo exchange refills the ghost rings. */
int cs_tile_histogram(cs_engine*, uint64_t* rows, uint64_t* cols);
size_t cs_tile_export(cs_engine*, void* records, size_t cap_records);
int cs_tile_retile(cs_engine*, uint32_t tile_cx0, uint32_t tile_cx1, uint32_t tile_cy0, uint32_t tile_cy1);
int cs_tile_import(cs_engine*, const void* records, size_t n);

/* Route followers (CS_HLP_ROUTE) on tiles.  Route numbers travel in halo records, so every tile's
 * route book must number routes alike.  Legs after the first start wherever an agent stands when it
 * reaches a waypoint (set_target, lib.rs:325-333 -> rmf/mod.rs:217-236): the step kernel answers
 * them from the book; a (start, goal) pair the book lacks is a MISS.  After cs_step the host collects
 * the misses of all tiles (cs_route_misses; with out == NULL just their number), sorts them by agent
 * id (the reference's canonical visiting order) and hands the merged list to EVERY tile
 * (cs_route_resolve): each plans the new routes in that order and assigns them to the agents it holds. */
typedef struct cs_route_miss {
  uint64_t id;   /* agent */
  uint32_t hlp;  /* planner handle */
  uint32_t slot; /* where the reporting tile holds the agent (opaque to other tiles) */
  double px, py; /* agent position: the start of the leg */
  double tx, ty; /* the waypoint: the goal of the leg */
} cs_route_miss;
size_t cs_route_misses(cs_engine*, cs_route_miss* out, size_t cap);
int cs_route_resolve(cs_engine*, const cs_route_miss* all, size_t n);

/* ---- tiles: the transport itself, over RCCL ------------------------------ */
/* The halo exchange and the spawn-flag all-reduce without any help from the host's runtime: the
 * engine binds RCCL (librccl.so.1: the copy already in the process, else the system's) at first
 * use and issues ncclSend / ncclRecv / ncclAllReduce on ITS stream, between cs_halo_pack_all and
 * cs_halo_unpack_all.  One communicator per engine: rank 0 calls cs_rccl_unique_id and hands the
 * 128 bytes to the other ranks by whatever means the host has (MPI, a file, torch.distributed),
 * every rank then calls cs_rccl_comm_init (collective, like ncclCommInitRank); or the host adopts a
 * communicator it already owns.  cs_halo_set_peers names the rank behind each direction
 * (CS_DIR_*; -1 = no neighbour).  Messages are the fixed-capacity buffers of cs_halo_set_buffers
 * (the record count travels in the header), so no size exchange precedes them. */
#define CS_RCCL_UNIQUE_ID_BYTES 128
int cs_rccl_unique_id(uint8_t* out_id);
int cs_rccl_comm_init(cs_engine*, int32_t n_ranks, int32_t rank, const uint8_t* id);
int cs_rccl_comm_adopt(cs_engine*, void* nccl_comm); /* an ncclComm_t of the host; not destroyed by the engine */
int cs_halo_set_peers(cs_engine*, const int32_t* peers8);
/* ncclGroupStart; ncclSend(send buffer) + ncclRecv(recv buffer) per direction with a peer;
 * ncclGroupEnd; all on the engine's stream.  With `axis` 0 / 1 only that pair of edge directions
 * (the two-phase schedule), with axis < 0 all eight. */
int cs_halo_exchange_rccl(cs_engine*, int32_t axis);
/* max over the ranks, element by element, in place (the OR of the spawn flags of
 * cs_spawn_probe_dev), on the engine's stream */
int cs_allreduce_max_i32_rccl(cs_engine*, int* values_dev, size_t n);
/* every rank's `bytes` bytes to every rank, in rank order (device buffers; recv_dev holds n_ranks * bytes), on the
 * engine's stream */
int cs_allgather_bytes_rccl(cs_engine*, const void* send_dev, void* recv_dev, size_t bytes);
/* One multi-GPU step of a tile in one call: cs_halo_pack_all -> cs_halo_exchange_rccl(-1) ->
 * cs_halo_unpack_all -> (with source-sinks: cs_spawn_probe_dev -> cs_allreduce_max_i32_rccl ->
 * cs_spawn_commit_dev on flags the engine keeps) -> cs_step, all on the engine's stream.  For hosts
 * without listeners, host planners or multi-leg route sinks (they use the split calls). */
int cs_tile_step_rccl(cs_engine*, double dt_seconds, cs_step_report* report);

/* ---- a crowd cut into spatial tiles behind one handle ------------------------------
 * The multi-tile form of `Simulation` (lib.rs:69-192): layout (tensor-product cuts, even or at the quantiles of a
 * set of positions), one tile engine per tile, halo buffers sized per direction, the exchange, the spawn flags OR-ed
 * over the tiles, route-cache misses, re-cuts and merged spatial queries.  Results equal the single engine's bit
 * for bit, the reference's clamp into row / column 0 along the domain's own low edges included
 * (location_hash_2d.rs:54-66); what a mesh does not reproduce is the alias of y beyond the row stride (Err here).
 *   rccl_unique_id null   every tile in this process on ONE device (exchanges are device copies on a shared stream)
 *   rccl_unique_id given  one tile per rank (n_ranks = tiles_x * tiles_y, also 1 x 1 with one rank), rank = tile index = tx * tiles_y + ty, halo records over RCCL from the
 *                  engine itself (ncclSend / ncclRecv for the halos, ncclAllReduce for the spawn flags,
 *                  ncclAllGather for what needs every rank's answer)
 *   host_transport given  the same one-tile-per-rank form over a transport of the host's (see above), instead of
 *                  RCCL or beside it (then it carries the gathers)
 * In both distributed forms every cs_mesh_* call is collective (every rank makes it, in the same order), and
 * cs_mesh_recut, cs_mesh_query_*_batch, cs_mesh_agent_count / cs_mesh_read_agents (the whole crowd on every
 * rank), cs_mesh_remove_agent and multi-leg route followers work across ranks.
 *                  Every rank passes the same grid and the same descriptor but for `rank` and the device: layout
 *                  and halo capacities are computed from them on each rank and are not exchanged.
 * cs_mesh_tile gives the underlying tile engines (profiling, snapshots, kernel statistics).
 * Failure: a step that fails on ANY tile (lib.rs:299-302: "Index out of bounds") stops the whole mesh: the step phase
 * still runs on every local tile, the ranks of a distributed mesh agree that somebody failed before anybody leaves
 * the schedule (after the step when the host waits for it anyway, i.e. with a report, listeners or host planners;
 * every 32 steps and in cs_mesh_synchronize when steps are made without waiting for the device), and from then on
 * every call on every rank returns the first error (the failing rank's own, "a tile of this mesh failed ..." elsewhere).
 * Between its failure and the agreed check a rank keeps issuing the collectives of its steps and nothing else. */
typedef struct cs_mesh cs_mesh;
/* A transport the HOST brings (MPI, gloo, sockets) for a distributed mesh: instead of RCCL (rccl_unique_id null:
 * the halo records and the spawn flags go through it, staged in pinned host memory), or beside it (both given:
 * RCCL moves halos and flags, this moves what the RCCL form leaves to the host: route-cache misses of multi-leg
 * sinks, re-cuts, merged queries).  Every buffer is host memory.  The calls are collective: every rank of the mesh
 * makes them in the same order (the mesh does, if every rank makes the same cs_mesh_* calls).  0 = success.
 *   exchange           n messages out and n in: send_host[k] (bytes[k] bytes) goes to rank peers[k], recv_host[k]
 *                      takes bytes[k] bytes from rank peers[k].  send_tags[k] is the direction (0..7) the message
 *                      leaves this rank in and travels with it; recv_tags[k] is the tag of the message expected
 *                      (the peer's direction towards this rank), should a pair of ranks exchange several.
 *   allreduce_max_i32  element-wise maximum over the ranks, in place
 *   allgather          every rank contributes `bytes` bytes (the same number everywhere); `all` receives
 *                      n_ranks * bytes, in rank order */
typedef int (*cs_mesh_exchange_fn)(void* user, size_t n, const int32_t* peers, const int32_t* send_tags,
                                   const int32_t* recv_tags, const void* const* send_host, void* const* recv_host,
                                   const size_t* bytes);
typedef int (*cs_mesh_allreduce_max_fn)(void* user, int32_t* values, size_t n);
typedef int (*cs_mesh_allgather_fn)(void* user, const void* mine, size_t bytes, void* all);
typedef struct cs_mesh_host_transport {
  void* user;
  cs_mesh_exchange_fn exchange;
  cs_mesh_allreduce_max_fn allreduce_max_i32;
  cs_mesh_allgather_fn allgather;
} cs_mesh_host_transport;
typedef struct cs_mesh_desc {
  uint32_t tiles_x, tiles_y;   /* 4 x 2 on 8 GPUs (BASELINE.json configs[2]); x = the index location_to_index
                                * multiplies by the row stride (location_hash_2d.rs:59) */
  uint32_t halo_cells;         /* >= ceil(largest eyesight / cell size) */
  uint32_t flags;              /* CS_CFG_* for every tile engine */
  int32_t device_ordinal;      /* the HIP device of this process's tile(s) */
  int32_t rank, n_ranks;       /* of the distributed form (rccl_unique_id given) */
  double density_per_cell;     /* expected agents per cell: sizes the halo buffers (0 = 16) */
  uint64_t capacity_hint;      /* per tile */
  const double* weights_xy;    /* optional: n_weights positions; the cuts go to the quantiles of their */
  size_t n_weights;            /*   row / column histograms (a clustered crowd, configs[4]) */
  const uint8_t* rccl_unique_id; /* distributed form: CS_RCCL_UNIQUE_ID_BYTES from cs_rccl_unique_id on one rank */
  const cs_mesh_host_transport* host_transport; /* distributed form without RCCL, or beside it (copied at creation) */
} cs_mesh_desc;
cs_mesh* cs_mesh_create(const cs_grid_desc* grid, const cs_mesh_desc* desc);
void cs_mesh_destroy(cs_mesh*);
const char* cs_mesh_last_error(const cs_mesh*); /* null mesh: why the calling thread's last cs_mesh_create failed */
size_t cs_mesh_local_tiles(const cs_mesh*);
cs_engine* cs_mesh_tile(cs_mesh*, size_t local_index);
int cs_mesh_tile_rect(const cs_mesh*, size_t local_index, uint32_t* rect4 /* cx0, cx1, cy0, cy1 */);
uint32_t cs_mesh_register_zanlungo(cs_mesh*, const cs_zanlungo_params*);
uint32_t cs_mesh_register_no_local_plan(cs_mesh*);
uint32_t cs_mesh_register_hlp(cs_mesh*, const cs_hlp_desc*);
/* a LocalPlanner that is host code (local_planner.rs:7-18; cs_register_lp_callback): every tile asks it for the agents
 * it owns, with their neighbours (ghosts of the tile included) in canonical order; such a mesh steps through the host
 * every step, like one with listeners */
uint32_t cs_mesh_register_lp_callback(cs_mesh*, cs_lp_batch_fn fn, void* user);
int cs_mesh_add_agents(cs_mesh*, const double* xy, size_t n, uint32_t hlp, uint32_t lp, double eyesight,
                       uint64_t* out_ids);                                        /* lib.rs:119-156 */
uint32_t cs_mesh_add_source_sink(cs_mesh*, const cs_source_sink_desc*);           /* lib.rs:159 */
void cs_mesh_remove_source_sink(cs_mesh*, uint32_t handle);                       /* lib.rs:164 */
int cs_mesh_remove_agent(cs_mesh*, uint64_t id);                                  /* lib.rs:176-192 */
void cs_mesh_event_recording(cs_mesh*, int on);
size_t cs_mesh_drain_events(cs_mesh*, cs_event* out, size_t cap);
int cs_mesh_step(cs_mesh*, double dt_seconds, cs_step_report* report);            /* lib.rs:195-383 */
int cs_mesh_synchronize(cs_mesh*);
size_t cs_mesh_agent_count(cs_mesh*);      /* the whole crowd (distributed forms: collective) */
size_t cs_mesh_read_agents(cs_mesh*, cs_agent_view* out, size_t cap);             /* ascending id; SIZE_MAX on error */
int cs_mesh_tile_counts(cs_mesh*, uint64_t* out_per_local_tile);
/* bytes the local tiles send per halo exchange, i.e. per step (fixed-capacity buffers: independent of the crowd) */
uint64_t cs_mesh_exchange_bytes(const cs_mesh*);
int cs_mesh_recut(cs_mesh*);
int cs_mesh_query_radius_batch(cs_mesh*, size_t n, const double* xy, const double* radius, size_t cap_per_query,
                               uint64_t* out_ids, uint64_t* out_counts);          /* spatial_index.rs:4-14 */
int cs_mesh_query_knn_batch(cs_mesh*, size_t n, const double* xy, size_t k, uint64_t* out_ids, uint64_t* out_counts);

#ifdef __cplusplus
}
#endif
#endif /* CROWDSTEP_H */
