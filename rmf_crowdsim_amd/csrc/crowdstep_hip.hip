// crowdstep_hip.hip — MI355X (gfx950) engine behind include/crowdstep.h.
//
// The per-step hot path of rmf_crowdsim's Simulation::step (reference
// rmf_crowdsim/src/lib.rs:195-383) as HIP kernels over cell-sorted SoA agent
// state.  See DESIGN.md for the data layout and the per-kernel rooflines.
//
// One translation unit, in parts (all in this directory):
//   cs_device_types.hip.inc   device tables; TTC / Zanlungo force arithmetic; re-binning
//   cs_kernels_sort.hip.inc   k_count, k_scan_*, k_scatter: the per-step cell re-sort
//                             (location_hash_2d.rs:126-149; implicit in Vec<HashSet>, :15)
//   cs_kernels_step.hip.inc   k_step_tiled / k_step_gather: neighbour query + Zanlungo +
//                             integrate + re-bin + waypoint/sink test
//                             (location_hash_2d.rs:240-258, zanlungo.rs:49-217, lib.rs:259-359)
//   cs_kernels_aux.hip.inc    k_halo_pack/unpack (tiles), k_spawn (lib.rs:199-254), radius query
//   cs_engine.hip.inc         the host engine: step state machine, tables, tiles, events
//   cs_rccl.hip.inc           RCCL bound at first use (halo transport of a tile)
//   this file                 includes + the extern "C" boundary
//
// Device state is f32 and CELL-RELATIVE: an agent is (stored cell, offset from
// that cell's origin), so relative positions between neighbours keep ~1e-7 m
// resolution at any domain size.  f64 appears only at the ABI.

#include <hip/hip_runtime.h>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <array>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "crowdstep.h"

#define CS_INVALID_CELL 0xFFFFFFFFu
#define CS_MAX_GROUPS 1048575u  // the group index travels in 16 or 20 bits of `meta` (GridDev::grp_bits)
#define CS_SPAWN_OCCUPANCY_RADIUS 0.4  // hard-coded in the reference, lib.rs:212-214
// Device ids are below 2^31 (`usize` in the reference): the tiled kernel reads "the neighbour's id is larger"
// (right_of_way_vel, zanlungo.rs:173-198 with priority = id) off the sign of a 32-bit difference.
#define CS_ID_LIMIT 0x7FFFFFFFull

#include "cs_device_types.hip.inc"
#include "cs_kernels_sort.hip.inc"
#include "cs_kernels_step.hip.inc"
#include "cs_kernels_aux.hip.inc"
#include "cs_engine.hip.inc"
#include "cs_rccl.hip.inc"

#define HIP_OK_E(e, call)                                                                     \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (e)->error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #call;         \
      return 90;                                                                              \
    }                                                                                         \
  } while (0)

// ===========================================================================
// C ABI (include/crowdstep.h)
// ===========================================================================
extern "C" {

uint32_t cs_abi_version(void) { return CS_ABI_VERSION; }

void cs_destroy(cs_engine* e) {
  if (!e) return;
  hipSetDevice(e->device);
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->aux_stream) hipStreamSynchronize(e->aux_stream);  // (an exchange made ahead may still be in flight there)
  if (e->rccl_comm && e->rccl_comm_owned && rccl_api::api().comm_destroy) rccl_api::api().comm_destroy(e->rccl_comm);
  e->free_arrays(e->buf[0]);
  e->free_arrays(e->buf[1]);
  hipFree(e->pref); hipFree(e->cell_count); hipFree(e->cell_start); hipFree(e->block_totals);
  hipFree(e->ctr); hipHostFree(e->ctr_host); hipFree(e->epi_dev); delete[] e->epi_host; hipFree(e->destroyed); hipFree(e->wp_events);
  for (auto& sn : e->snap) {
    if (sn.in_flight) hipEventSynchronize(sn.copied);
    hipFree(sn.dev); hipHostFree(sn.host); hipFree(sn.count_dev); hipHostFree(sn.count_host);
    if (sn.gathered) hipEventDestroy(sn.gathered);
    if (sn.copied) hipEventDestroy(sn.copied);
  }
  if (e->copy_stream) hipStreamDestroy(e->copy_stream);
  if (e->peek_host) hipHostFree(e->peek_host);
  if (e->aux_stream) {
    hipStreamSynchronize(e->aux_stream);
    hipStreamDestroy(e->aux_stream);
    hipEventDestroy(e->ev_sorted); hipEventDestroy(e->ev_border); hipEventDestroy(e->ev_xchg);
  }
  hipFree(e->route_desc_dev); hipFree(e->route_xy_dev); hipFree(e->route_book_dev); hipFree(e->hlp_scale_dev); hipFree(e->route_pending_dev);
  hipFree(e->groups_dev); hipFree(e->sinks_dev); hipFree(e->waypoints_dev);
  hipFree(e->src_cell_start); hipFree(e->src_sorted); hipFree(e->src_occupied);
  hipFree(e->want_dev); hipFree(e->spawned_slots_dev); hipFree(e->spawn_scratch); hipHostFree(e->want_host); hipFree(e->blk_desc); hipFree(e->blk_desc_back); hipFree(e->n_blocks_dev); hipFree(e->n_blocks_back); hipFree(e->band_prefix); hipFree(e->tile_spill); hipFree(e->spawn_rec_dev); hipFree(e->find_dev); hipFree(e->step_flags_dev); hipFree(e->query_scratch); hipFree(e->scan_tile_sums);
  for (auto& t : e->timed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
  for (auto ev : e->event_pool) hipEventDestroy(ev);
  if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
  delete e;
}

// why the last cs_create of this thread returned null (cs_last_error(nullptr))
static thread_local std::string g_create_error;
static cs_engine* create_failed(cs_engine* e, const std::string& why) {
  g_create_error = why;
  fprintf(stderr, "crowdstep: %s\n", why.c_str());
  if (e) cs_destroy(e);
  (void)hipGetLastError();  // the failure is reported here: leave no stale error for the caller's next HIP call
  return nullptr;
}

// Simulation::new(LocationHash2D::new(..)), lib.rs:103 + location_hash_2d.rs:33-51
cs_engine* cs_create(const cs_grid_desc* grid, const cs_device_cfg* cfg) {
  if (!grid) return create_failed(nullptr, "null grid description");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return create_failed(nullptr, "no HIP device visible; the engine has no CPU fallback");
  cs_engine* e = new cs_engine();
  e->grid = *grid;
  e->device = cfg ? cfg->device_ordinal : 0;
  e->flags = cfg ? cfg->flags : 0;
  if (e->device < 0 || e->device >= ndev || hipSetDevice(e->device) != hipSuccess) {
    const int ordinal = e->device;
    delete e;  // nothing allocated yet (and cs_destroy would select the device again)
    return create_failed(nullptr, "no HIP device with ordinal " + std::to_string(ordinal));
  }
  // (width / cell) as usize is the row stride, used on BOTH axes, and the number of x rows
  // that fit is len / stride = (height / cell) as usize (location_hash_2d.rs:36-37,59)
  e->gnx = sat_usize(grid->width / grid->cell_size);
  e->gny = sat_usize(grid->height / grid->cell_size);
  unsigned __int128 gnc = (unsigned __int128)e->gnx * e->gny;
  if (gnc >= 0x7FFFFFFFull || e->gnx >= 0x7FFFFFFFull)
    return create_failed(e, "grid too large for 32-bit cell indices (" + std::to_string(e->gnx) + " x " +
                                std::to_string(e->gny) + " cells)");
  std::memset(&e->gdev, 0, sizeof e->gdev);
  e->gdev.cs = (float)grid->cell_size;
  e->gdev.cs_lo = (float)(grid->cell_size - (double)e->gdev.cs);
  e->gdev.inv_cs = 1.0f / e->gdev.cs;
  {  // fixed-point unit of the neighbour pass (fix_rel): the cell size is at most 2^23 units, i.e.
     // 2^-22 m for cells of (1, 2] m (two f32 spacings of an offset in the upper half of the
     // cell), so that a window of up to ~120 columns fits 2^30 units (tile_max_cols)
    int bits = 23 - (int)std::ceil(std::log2(std::max(grid->cell_size, 1e-30)));
    bits = std::max(-60, std::min(bits, 60));
    e->gdev.fix_scale = (float)std::ldexp(1.0, bits);
    e->gdev.fix_inv = (float)std::ldexp(1.0, -bits);
    e->gdev.cs_fix = (int32_t)std::llrint(grid->cell_size * std::ldexp(1.0, bits));
  }
  e->gdev.grp_bits = 16u;  // (cs_engine::room_for_group widens it)
  if (cfg && cfg->stream) {
    e->stream = (hipStream_t)cfg->stream;
  } else {
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
      return create_failed(e, "hipStreamCreate failed");
    e->own_stream = true;
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, e->device);
  e->backend = std::string("hip:") + prop.gcnArchName;
  {
    const void* kernels[6] = {reinterpret_cast<const void*>(k_step_tiled<false, false>), reinterpret_cast<const void*>(k_step_tiled<true, false>),
                              reinterpret_cast<const void*>(k_step_tiled<false, true>), reinterpret_cast<const void*>(k_step_tiled<true, true>),
                              reinterpret_cast<const void*>(k_step_tiled<false, false, true>),
                              reinterpret_cast<const void*>(k_step_tiled<true, false, true>)};
    for (int k = 0; k < 6; ++k) {
      if (hipFuncSetAttribute(kernels[k], hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
        (void)hipGetLastError();  // not fatal: the default 64 KiB covers the usual staged tile (wide cells need more)
      // static LDS of the tiled kernel: part of a workgroup's share of the CU's 160 KiB (the instantiations with builder
      // workgroups, k >= 2, carry the window builder's small arrays as well: priced for themselves)
      hipFuncAttributes fa;
      uint32_t& slot = (k < 2 || k >= 4) ? e->tile_static_lds : e->tile_static_lds_builders;
      if (hipFuncGetAttributes(&fa, kernels[k]) == hipSuccess && fa.sharedSizeBytes)
        slot = std::max(slot == 512u ? 0u : slot, (uint32_t)fa.sharedSizeBytes);
      (void)hipGetLastError();
    }
  }
  if (const char* v = getenv("CS_WINDOWS_KEEP")) e->windows_shadow = atoi(v) != 0;  // 0: the window builder in every step's own launch
  if (const char* v = getenv("CS_WINDOWS_SLACK")) e->windows_slack_pct = (uint32_t)std::max(0, atoi(v));  // room in a kept window, percent
  if (const char* v = getenv("CS_WINDOWS_KEEP_MAX_AGENTS")) e->windows_keep_max_agents = (uint32_t)std::max(0, atoi(v));
  if (const char* v = getenv("CS_TILE_BLOCKS_PER_CU")) e->tile_blocks_per_cu = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_TILE_MIN_ROWS")) e->tile_min_rows = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_TILE_LIST_CAP")) e->tile_list_cap = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_TILE_SPILL_ROWS")) e->tile_spill_rows = atoi(v);
  if (const char* v = getenv("CS_TILE_AGENTS_SLACK")) e->tile_agents_slack = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_HALO_FUSE")) e->halo_fuse = atoi(v) != 0;
  if (const char* v = getenv("CS_TILE_ASYNC")) e->kTileAsync = (uint32_t)std::max(1, atoi(v));
  if (const char* v = getenv("CS_TILE_SPLIT")) e->tile_split_force = atoi(v) != 0;
  if (const char* v = getenv("CS_OVERLAP_MODE")) e->overlap_early = v[0] != 's';  // "split": rounds 2-4's two launches
  if (const char* v = getenv("CS_DEBUG_CTX_BY_VALUE")) e->debug_ctx_by_value = atoi(v) != 0;  // (measurement only)
  if (const char* v = getenv("CS_DEBUG_SORT_EVERY")) e->debug_sort_every = (uint32_t)std::max(0, atoi(v));  // (measurement only)
  if (const char* v = getenv("CS_TILE_STAGE_CAP")) e->tile_stage_cap = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_TILE_WINDOWS_CAP")) e->tile_windows_cap = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_CHECK_WINDOWS")) e->check_windows = atoi(v) != 0;
  if (const char* v = getenv("CS_SCAN_ONEPASS")) e->scan_onepass = atoi(v) != 0;
  if (const char* v = getenv("CS_TILE_ROWS")) e->tile_rows = std::min<uint32_t>(TILE_MAX_OWN_ROWS, std::max(1, atoi(v)));
  if (const char* v = getenv("CS_TILE_TARGET")) e->tile_target = std::min<uint32_t>(4 * TILE_THREADS, std::max(32, atoi(v)));
  const bool is_tile = cfg && (cfg->tile_cx1 | cfg->tile_cy1);
  if (e->set_geometry(is_tile, is_tile ? cfg->tile_cx0 : 0, is_tile ? cfg->tile_cx1 : 0, is_tile ? cfg->tile_cy0 : 0,
                      is_tile ? cfg->tile_cy1 : 0, is_tile ? cfg->halo_cells : 0) != 0)
    return create_failed(e, e->error);
  bool ok = true;
  ok = ok && hipMalloc(&e->ctr, sizeof(Counters)) == hipSuccess;
  ok = ok && hipHostMalloc(&e->ctr_host, sizeof(Counters)) == hipSuccess;
  if (ok) hipMemset(e->ctr, 0, sizeof(Counters));
  if (const char* v = getenv("CS_FIRST_AGENT_ID")) {  // test knob: ids near the 31-bit limit without 2^31 agents
    e->next_id = std::min<uint64_t>((uint64_t)strtoull(v, nullptr, 10), CS_ID_LIMIT);
    const uint32_t nid = (uint32_t)e->next_id;
    if (ok) hipMemcpy(&e->ctr->next_id, &nid, sizeof nid, hipMemcpyHostToDevice);
  }
  if (!ok || e->upload_sinks() != 0 ||
      e->reserve(cfg && cfg->capacity_hint ? cfg->capacity_hint : 1024) != 0) {
    return create_failed(e, "device allocation failed: " + e->error);
  }
  return e;
}

const char* cs_last_error(const cs_engine* e) { return e ? e->error.c_str() : g_create_error.c_str(); }
const char* cs_backend_name(const cs_engine* e) { return e ? e->backend.c_str() : ""; }

uint32_t cs_register_zanlungo(cs_engine* e, const cs_zanlungo_params* p) {
  e->lp_params.push_back(*p);
  e->lp_kinds.push_back(1u);
  e->lp_callbacks.emplace_back();
  return (uint32_t)e->lp_kinds.size() - 1;
}
uint32_t cs_register_no_local_plan(cs_engine* e) {
  cs_zanlungo_params z;
  std::memset(&z, 0, sizeof z);
  z.agent_mass = 1.0;
  z.force_distance = 1.0;
  e->lp_params.push_back(z);
  e->lp_kinds.push_back(0u);
  e->lp_callbacks.emplace_back();
  return (uint32_t)e->lp_kinds.size() - 1;
}
uint32_t cs_register_lp_callback(cs_engine* e, cs_lp_batch_fn fn, void* user) {
  if (!fn) {
    e->error = "cs_register_lp_callback: the planner's function is null";
    return UINT32_MAX;
  }
  cs_zanlungo_params z;
  std::memset(&z, 0, sizeof z);
  z.agent_mass = 1.0;
  z.force_distance = 1.0;
  e->lp_params.push_back(z);
  e->lp_kinds.push_back(2u);
  cs_engine::LpCallback cb;
  cb.fn = fn;
  cb.user = user;
  e->lp_callbacks.push_back(cb);
  return (uint32_t)e->lp_kinds.size() - 1;
}
uint32_t cs_register_hlp(cs_engine* e, const cs_hlp_desc* d) {
  if (d->kind == CS_HLP_ROUTE) {
    if (!d->route_plan || !(d->route_scale > 0.0) || !(d->route_arrive >= 0.0)) {
      e->error = "route planner: route_plan, route_scale > 0 and route_arrive >= 0 are required";
      return UINT32_MAX;
    }
  }
  e->hlps.push_back(*d);
  return (uint32_t)e->hlps.size() - 1;
}

int cs_add_agents(cs_engine* e, const double* xy, size_t n, uint32_t hlp, uint32_t lp,
                  double eyesight, uint64_t* out_ids) {
  hipSetDevice(e->device);
  if (hlp >= e->hlps.size() || lp >= e->lp_kinds.size()) {
    e->error = "unknown planner handle";
    return 2;
  }
  // (a group this call only REUSES needs no room: asking first would widen the meta word, or refuse the call, for nothing)
  if (e->plain_groups.find(std::make_tuple(hlp, lp, eyesight)) == e->plain_groups.end())
    if (int rc = e->room_for_group(0)) return rc;
  uint32_t g = e->make_group(hlp, lp, eyesight, -1);
  return e->add_agents(xy, n, g, UINT32_MAX, out_ids);
}

// Simulation::remove_agents, lib.rs:176-192.  0 = removed, 2 = nobody here has this id (on a tile:
// another tile may), anything else = the engine's own failure (poisoned, HIP error).
int cs_remove_agent(cs_engine* e, uint64_t id) {
  hipSetDevice(e->device);
  e->halo_invalidate();  // also where the agent is not found: every tile is asked, all must agree on what follows
  for (size_t k = 0; k < e->limbo.size(); ++k)
    if (e->limbo[k].id == id) {  // an agent the index never took (lib.rs:133-149) is removed like any other (:176-192)
      const HostGroup& g = e->groups[e->limbo[k].group];
      const cs_hlp_desc& p = e->hlps[g.hlp];
      if (p.kind == CS_HLP_CALLBACK && p.remove_agent) p.remove_agent(p.user, id);
      e->limbo.erase(e->limbo.begin() + (long)k);
      cs_event ev;
      ev.kind = CS_EVENT_DESTROYED;
      ev.source_sink = UINT32_MAX;
      ev.id = id;
      ev.x = ev.y = 0;
      if (e->record_events) e->events.push_back(ev);
      return 0;
    }
  if (int rc = e->refresh_counts()) return rc;
  uint32_t found[2] = {0xFFFFFFFFu, 0u};
  if (id < 0xFFFFFFFFull && e->n_slots) {
    if (!e->find_dev && hipMalloc(&e->find_dev, 2 * sizeof(uint32_t)) != hipSuccess) {
      e->error = "HIP error while removing an agent";
      return 90;
    }
    bool ok = hipMemcpyAsync(e->find_dev, found, sizeof found, hipMemcpyHostToDevice, e->stream) == hipSuccess;
    hipLaunchKernelGGL(k_remove_by_id, dim3((e->n_slots + 255u) / 256u), dim3(256), 0, e->stream, e->buf[e->cur],
                       e->n_slots, e->ctr, e->gdev.tile, (uint32_t)id, e->find_dev);
    ok = ok && hipMemcpyAsync(found, e->find_dev, sizeof found, hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
         hipStreamSynchronize(e->stream) == hipSuccess;
    if (!ok) {
      e->error = "HIP error while removing an agent";
      return 90;
    }
  }
  if (found[0] == 0xFFFFFFFFu) {
    e->error = "unknown agent id";
    return 2;
  }
  const HostGroup& g = e->groups[meta_group(e->gdev, found[1])];
  const cs_hlp_desc& p = e->hlps[g.hlp];
  if (p.kind == CS_HLP_CALLBACK && p.remove_agent) p.remove_agent(p.user, id);
  e->sorted = false;
  e->hist_valid = false;
  e->occ_valid = false;
  e->halo_invalidate();
  e->n_alive_host -= 1;
  cs_event ev;
  ev.kind = CS_EVENT_DESTROYED;
  ev.source_sink = g.sink >= 0 ? (uint32_t)g.sink : UINT32_MAX;
  ev.id = id;
  ev.x = ev.y = 0;
  if (e->record_events) e->events.push_back(ev);
  return 0;
}

size_t cs_source_sink_slots(cs_engine* e) { return e->sinks.size(); }
uint64_t cs_device_bytes(cs_engine* e) { return e->device_bytes(); }

uint64_t cs_kernel_stat(cs_engine* e, uint32_t which) {
  hipSetDevice(e->device);
  if (which == CS_STAT_EXCHANGES_AHEAD) return e->n_exchanges_ahead;
  if (which == CS_STAT_EXCHANGES_AHEAD_USED) return e->n_exchanges_ahead_used;
  if (which == CS_STAT_STEPS_ON_KEPT_WINDOWS) return e->n_steps_on_kept_windows;
  Counters c;
  if (e->read_counters(&c)) return 0;
  switch (which) {
    case CS_STAT_WINDOWS_OFF_LDS: return c.n_win_off_lds;
    case CS_STAT_WINDOWS_CHUNKED: return c.n_win_chunked;
#ifdef CS_TILE_TRIPS
    case 100: case 101: case 102: case 103: case 104: case 105: case 106: case 107: case 108: case 109: case 110: case 111:
      return c.dbg[which - 100];
#endif
    default: return 0;
  }
}

uint32_t cs_add_source_sink(cs_engine* e, const cs_source_sink_desc* d) {
  if (d->n_waypoints == 0) {
    e->error = "a source-sink needs at least one waypoint";
    return UINT32_MAX;
  }
  hipSetDevice(e->device);
  if (e->room_for_group(d->n_waypoints)) return UINT32_MAX;
  HostSink s;
  s.d = *d;
  s.waypoints.assign(d->waypoints_xy, d->waypoints_xy + 2 * d->n_waypoints);
  s.d.waypoints_xy = nullptr;
  uint32_t handle = (uint32_t)e->sinks.size();
  if (e->tile && d->hlp < e->hlps.size() && e->hlps[d->hlp].kind == CS_HLP_ROUTE) {
    // A tile cannot ask its host for a route in the middle of a step and stay in step with its
    // neighbours: the first leg's route is put into the book here, in sink order, the same on
    // every tile (halo records carry route numbers).  Later legs start wherever an agent stands:
    // they are answered from the book on the device, and what misses it comes back through
    // cs_route_misses / cs_route_resolve after the step, merged over all tiles by the host.
    s.spawn_route = e->route_lookup(d->hlp, d->source_x, d->source_y, s.waypoints[0], s.waypoints[1]);
    if (!s.spawn_route) {
      e->error = "route planner found no route for this source-sink (tile engines plan at registration)";
      return UINT32_MAX;
    }
  }
  s.group = e->make_group(d->hlp, d->lp, d->agent_eyesight_range, (int32_t)handle);
  e->sinks.push_back(s);
  e->sinks_dirty = true;
  return handle;
}

void cs_remove_source_sink(cs_engine* e, uint32_t handle) {
  // lib.rs:164-168: only the registry entry goes.  Agents it spawned keep walking
  // (the reference would panic on their next waypoint lookup, lib.rs:309; here they
  // simply stop being tested).
  if (handle < e->sinks.size() && e->sinks[handle].alive) {
    e->sinks[handle].alive = false;
    e->sinks_dirty = true;
  }
}

int cs_step(cs_engine* e, double dt_seconds, cs_step_report* report) {
  hipSetDevice(e->device);
  const int rc = e->step(dt_seconds, report);
  if (rc == 0) e->steps_done += 1;
  return rc;
}

int cs_snapshot_request(cs_engine* e) {
  hipSetDevice(e->device);
  return e->snapshot_request();
}

int cs_snapshot_acquire(cs_engine* e, int wait, const cs_snapshot_record** out, size_t* n,
                        uint64_t* step_index) {
  hipSetDevice(e->device);
  return e->snapshot_acquire(wait, out, n, step_index);
}

int cs_synchronize(cs_engine* e) {
  hipSetDevice(e->device);
  if (int rc = e->refresh_counts()) return rc;
  Counters c;
  if (int rc = e->read_counters(&c)) return rc;
  e->prof_collect();
  if (c.n_out_of_bounds) {
    e->poisoned = true;
    e->error = "Index out of bounds";
    return 1;
  }
  return 0;
}

size_t cs_agent_count(cs_engine* e) {
  hipSetDevice(e->device);
  e->refresh_counts();
  return (size_t)e->n_alive_host + e->limbo.size();
}

size_t cs_read_agents(cs_engine* e, cs_agent_view* out, size_t cap) {
  hipSetDevice(e->device);
  if (e->refresh_counts() != 0) return 0;
  cs_engine::HostState h;
  if (e->download(&h) != 0) return 0;
  std::vector<uint32_t> live;
  for (uint32_t i = 0; i < e->n_slots; ++i)
    if (h.cell[i] != CS_INVALID_CELL) live.push_back(i);
  std::sort(live.begin(), live.end(), [&](uint32_t a, uint32_t b) { return h.id[a] < h.id[b]; });
  size_t n = std::min(cap, live.size());
  for (size_t k = 0; k < n; ++k) {
    uint32_t i = live[k];
    out[k].id = h.id[i];
    e->to_global(h.cell[i], h.off[i].x, h.off[i].y, &out[k].x, &out[k].y);
    out[k].vx = h.vel[i].x;
    out[k].vy = h.vel[i].y;
    out[k].next_waypoint = meta_waypoint(e->gdev, h.meta[i]);
    out[k].eyesight_range = e->groups[meta_group(e->gdev, h.meta[i])].eyesight;
  }
  if (!e->limbo.empty()) {  // the agents the index never took: as created (lib.rs:133-144), in id order with the rest
    for (const cs_engine::LimboAgent& l : e->limbo) {
      if (n >= cap) break;
      cs_agent_view v;
      std::memset(&v, 0, sizeof v);
      v.id = l.id;
      v.x = l.x;
      v.y = l.y;
      v.eyesight_range = e->groups[l.group].eyesight;
      out[n++] = v;
    }
    std::sort(out, out + n, [](const cs_agent_view& a, const cs_agent_view& b) { return a.id < b.id; });
  }
  return n;
}

size_t cs_drain_events(cs_engine* e, cs_event* out, size_t cap) {
  size_t n = std::min(cap, e->events.size());
  for (size_t i = 0; i < n; ++i) out[i] = e->events[i];
  e->events.erase(e->events.begin(), e->events.begin() + n);
  return n;
}

void cs_event_recording(cs_engine* e, int on) {
  hipSetDevice(e->device);
  e->refresh_counts();
  e->record_events = on != 0;
  if (!on) e->events.clear();
}

// Radius query on the device index: ids in reference cell order (x-major, y-minor, ascending id
// inside a cell) and their squared distances.  Returns the full count.
// `every_alias`: walk the whole rectangle the reference walks.  A rectangle wider than the grid
// reaches a cell of row R once through every row <= R (y runs past the row stride into the next
// rows, location_hash_2d.rs:74-85), and the reference lists its members every time; the public
// query reproduces that, the k-NN search (which removes duplicates anyway) visits at most two
// rows' worth of y.
static size_t radius_query(cs_engine* e, double radius, double x, double y, std::vector<uint32_t>* ids,
                           std::vector<float>* d2, size_t cap, bool every_alias) {
  if (e->tile) {
    e->error = "spatial queries are not available on a tile engine";
    return 0;
  }
  if (e->refresh_counts() != 0) return 0;
  if (e->ensure_index() != 0) return 0;
  // get_bounds (:103-122) in f64 on the global query point
  auto fl = [&](double v, double o) -> long long {
    double f = std::floor((v - o) / e->grid.cell_size);
    if (f != f) return 0;
    if (f > 4e18) return (long long)4e18;
    if (f < -4e18) return (long long)-4e18;
    return (long long)f;
  };
  long long lx = fl(x - radius, e->grid.offset_x), hx = fl(x + radius, e->grid.offset_x);
  long long ly = fl(y - radius, e->grid.offset_y), hy = fl(y + radius, e->grid.offset_y);
  lx = std::max(lx, -1ll); ly = std::max(ly, -1ll);
  hx = std::min(hx, (long long)(e->ncells / std::max<uint64_t>(e->nx, 1)));
  hy = std::min(hy, every_alias ? (long long)e->ncells : (long long)e->nx * 2);
  // query point relative to a reference cell: the cell of the point clamped into the grid
  long long qx = std::min(std::max(fl(x, e->grid.offset_x), 0ll), (long long)e->nx - 1);
  long long qy = std::min(std::max(fl(y, e->grid.offset_y), 0ll), (long long)e->nx - 1);
  float qox = (float)((x - e->grid.offset_x) - (double)qx * e->grid.cell_size);
  float qoy = (float)((y - e->grid.offset_y) - (double)qy * e->grid.cell_size);
  uint32_t qcap = (uint32_t)std::max<size_t>(1, std::min<size_t>(cap, 1u << 22));
  uint32_t* d_out = nullptr;
  float* d_d2 = nullptr;
  uint32_t* d_cnt = nullptr;
  bool ok = hipMalloc(&d_out, qcap * sizeof(uint32_t)) == hipSuccess &&
            hipMalloc(&d_d2, qcap * sizeof(float)) == hipSuccess &&
            hipMalloc(&d_cnt, sizeof(uint32_t)) == hipSuccess;
  uint32_t cnt = 0;
  if (ok) {
    hipLaunchKernelGGL(k_query_radius, dim3(1), dim3(64), 0, e->stream, e->gdev, e->buf[e->cur],
                       e->cell_start, lx, hx, ly, hy, (uint32_t)qx, (uint32_t)qy, qox, qoy, (float)radius,
                       d_out, d_d2, qcap, d_cnt);
    ok = hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
         hipStreamSynchronize(e->stream) == hipSuccess;
    uint32_t m = ok ? std::min(cnt, qcap) : 0;
    ids->resize(m);
    d2->resize(m);
    if (m) {
      ok = ok && hipMemcpy(ids->data(), d_out, m * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
      ok = ok && hipMemcpy(d2->data(), d_d2, m * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
    }
  }
  hipFree(d_out);
  hipFree(d_d2);
  hipFree(d_cnt);
  if (!ok) {
    e->error = "HIP error in a spatial query";
    ids->clear();
    d2->clear();
    return 0;
  }
  return cnt;
}

// The batch form of radius_query: n queries in one launch (k_query_radius_batch).  ids / d2 / cells
// hold cap entries per query, counts the full count per query.  On a tile engine the query runs
// against the local grid and reports owned agents only.
static int radius_query_batch(cs_engine* e, size_t n, const double* xy, const double* radius, size_t cap,
                              std::vector<uint32_t>* ids, std::vector<float>* d2, std::vector<uint32_t>* cells,
                              std::vector<uint32_t>* counts, bool every_alias, bool with_ghosts = false) {
  counts->assign(n, 0u);
  ids->assign(n * cap, 0u);
  d2->assign(n * cap, 0.0f);
  cells->assign(n * cap, 0u);
  if (n == 0 || cap == 0) return 0;
  if (n > 0x7FFFFFFFull || n * cap > 0x7FFFFFFFull) {
    e->error = "batch query too large";
    return 3;
  }
  if (int rc = e->refresh_counts()) return rc;
  if (int rc = e->ensure_index()) return rc;
  std::vector<QueryDev> q(n);
  const long long n_rows = (long long)(e->ncells / std::max<uint64_t>(e->nx, 1));
  auto fl = [&](double v, double o) -> long long {
    double f = std::floor((v - o) / e->grid.cell_size);
    if (f != f) return 0;
    if (f > 4e18) return (long long)4e18;
    if (f < -4e18) return (long long)-4e18;
    return (long long)f;
  };
  for (size_t k = 0; k < n; ++k) {
    const double x = xy[2 * k], y = xy[2 * k + 1], r = radius[k];
    // get_bounds (:103-122) in f64 on the global query point, then into the local grid of a tile
    const long long ox = (long long)e->gdev.org_x, oy = (long long)e->gdev.org_y;
    long long lx = fl(x - r, e->grid.offset_x) - ox, hx = fl(x + r, e->grid.offset_x) - ox;
    long long ly = fl(y - r, e->grid.offset_y) - oy, hy = fl(y + r, e->grid.offset_y) - oy;
    lx = std::max(lx, -1ll);
    ly = std::max(ly, -1ll);
    hx = std::min(hx, n_rows);
    hy = std::min(hy, e->tile ? (long long)e->nx - 1 : (every_alias ? (long long)e->ncells : (long long)e->nx * 2));
    // query point relative to a reference cell: the cell of the point clamped into the (local) grid
    const long long qx = std::min(std::max(fl(x, e->grid.offset_x) - ox, 0ll), std::max(n_rows - 1, 0ll));
    const long long qy = std::min(std::max(fl(y, e->grid.offset_y) - oy, 0ll), (long long)e->nx - 1);
    QueryDev& Q = q[k];
    Q.lx = lx; Q.hx = hx; Q.ly = ly; Q.hy = hy;
    Q.qcx = (uint32_t)qx;
    Q.qcy = (uint32_t)qy;
    Q.qox = (float)((x - e->grid.offset_x) - (double)(qx + ox) * e->grid.cell_size);
    Q.qoy = (float)((y - e->grid.offset_y) - (double)(qy + oy) * e->grid.cell_size);
    Q.r = (float)r;
    Q.pad = 0;
  }
  // device scratch of the batch, kept on the engine and grown as needed (a host local planner asks a batch per planner
  // and step: five hipMalloc / hipFree pairs per call used to cost more than the query itself)
  auto up = [](size_t b) { return (b + 255u) & ~(size_t)255u; };
  const size_t o_ids = up(n * sizeof(QueryDev)), o_d2 = o_ids + up(n * cap * sizeof(uint32_t)),
               o_cells = o_d2 + up(n * cap * sizeof(float)), o_cnt = o_cells + up(n * cap * sizeof(uint32_t)),
               total = o_cnt + up(n * sizeof(uint32_t));
  bool ok = true;
  if (total > e->query_scratch_bytes) {
    hipStreamSynchronize(e->stream);
    hipFree(e->query_scratch);
    e->query_scratch = nullptr;
    e->query_scratch_bytes = 0;
    ok = hipMalloc(&e->query_scratch, total + total / 4) == hipSuccess;
    if (ok) e->query_scratch_bytes = total + total / 4;
  }
  if (ok) {
    unsigned char* base = static_cast<unsigned char*>(e->query_scratch);
    QueryDev* d_q = reinterpret_cast<QueryDev*>(base);
    uint32_t* d_ids = reinterpret_cast<uint32_t*>(base + o_ids);
    float* d_d2 = reinterpret_cast<float*>(base + o_d2);
    uint32_t* d_cells = reinterpret_cast<uint32_t*>(base + o_cells);
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(base + o_cnt);
    ok = hipMemcpyAsync(d_q, q.data(), n * sizeof(QueryDev), hipMemcpyHostToDevice, e->stream) == hipSuccess;
    hipLaunchKernelGGL(k_query_radius_batch, dim3((uint32_t)n), dim3(64), 0, e->stream, e->gdev, e->buf[e->cur],
                       e->cell_start, d_q, (uint32_t)n, (uint32_t)e->gnx, (e->tile && e->ghosts_present && !with_ghosts) ? 1u : 0u, d_ids,
                       d_d2, d_cells, (uint32_t)cap, d_cnt);
    ok = ok && hipMemcpyAsync(counts->data(), d_cnt, n * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
         hipMemcpyAsync(ids->data(), d_ids, n * cap * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
         hipMemcpyAsync(d2->data(), d_d2, n * cap * sizeof(float), hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
         hipMemcpyAsync(cells->data(), d_cells, n * cap * sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
         hipStreamSynchronize(e->stream) == hipSuccess;
  }
  if (!ok) {
    e->error = "HIP error in a batch of spatial queries";
    return 90;
  }
  return 0;
}

int cs_query_radius_batch(cs_engine* e, size_t n, const double* xy, const double* radius, size_t cap_per_query,
                          uint64_t* out_ids, uint64_t* out_counts, float* out_d2, uint32_t* out_cells) {
  hipSetDevice(e->device);
  std::vector<uint32_t> ids, cells, counts;
  std::vector<float> d2;
  if (int rc = radius_query_batch(e, n, xy, radius, cap_per_query, &ids, &d2, &cells, &counts, true)) return rc;
  for (size_t k = 0; k < n; ++k) {
    if (out_counts) out_counts[k] = counts[k];
    const size_t m = std::min<size_t>(counts[k], cap_per_query);
    for (size_t i = 0; i < m; ++i) {
      out_ids[k * cap_per_query + i] = ids[k * cap_per_query + i];
      if (out_d2) out_d2[k * cap_per_query + i] = d2[k * cap_per_query + i];
      if (out_cells) out_cells[k * cap_per_query + i] = cells[k * cap_per_query + i];
    }
  }
  return 0;
}

// A step's turn of the host LocalPlanners (cs_register_lp_callback; local_planner.rs:7-18 called at lib.rs:276-291).
// Runs inside cs_step on the SORTED start-of-step state, after the spawn phase and the high-level planners'
// callbacks: (1) k_lp_recommended writes the velocity the high-level planner recommends for every agent of such a
// planner, (2) the state comes to the host, (3) the neighbours of those agents come from the device index through
// the batch radius query (cells x-major / y-minor, ascending id in a cell: the step's canonical order; the agent
// itself dropped, lib.rs:284; a tile's ghosts included), (4) every planner is called once with its agents in
// ascending id, (5) the answers go back by slot and the step kernels read them instead of evaluating a planner.
static int lp_callbacks_eval(cs_engine* e, const StepParams& P, const EpilogueCtx* Ep) {
  const uint32_t n = e->n_slots;
  hipLaunchKernelGGL(k_lp_recommended, dim3((n + 255u) / 256u), dim3(256), 0, e->stream, P, e->view(e->cur), Ep, e->pref,
                     e->lp_vel);
  cs_engine::HostState h;
  if (int rc = e->download(&h)) return rc;  // (synchronises the stream)
  std::vector<float2> rec(n);
  if (hipMemcpy(rec.data(), e->lp_vel, (size_t)n * sizeof(float2), hipMemcpyDeviceToHost) != hipSuccess) {
    e->error = "HIP error in a host local planner's turn";
    return 90;
  }
  // the agents of host planners that THIS engine steps (a tile: the owned ones), by planner, ascending id
  std::vector<std::vector<uint32_t>> by_lp(e->lp_kinds.size());
  for (uint32_t i = 0; i < n; ++i) {
    if (h.cell[i] == CS_INVALID_CELL) continue;
    const auto& g = e->groups[meta_group(e->gdev, h.meta[i])];
    if (e->lp_kinds[g.lp] != 2u) continue;
    if (e->tile) {
      const uint32_t cx = h.cell[i] / (uint32_t)e->nx, cy = h.cell[i] - cx * (uint32_t)e->nx;
      if (cx < e->gdev.own_x0 || cx >= e->gdev.own_x1 || cy < e->gdev.own_y0 || cy >= e->gdev.own_y1) continue;
    }
    by_lp[g.lp].push_back(i);
  }
  // id -> slot, for the neighbours' state: a sorted table of the live agents (ids grow without bound in a crowd fed
  // by source-sinks: a table indexed by id would be as large as the largest id ever handed out)
  std::vector<std::pair<uint32_t, uint32_t>> slot_of_id;
  slot_of_id.reserve(n);
  for (uint32_t i = 0; i < n; ++i)
    if (h.cell[i] != CS_INVALID_CELL) slot_of_id.emplace_back(h.id[i], i);
  std::sort(slot_of_id.begin(), slot_of_id.end());
  auto find_slot = [&](uint32_t id, uint32_t* slot) {
    const auto it = std::lower_bound(slot_of_id.begin(), slot_of_id.end(), std::make_pair(id, 0u));
    if (it == slot_of_id.end() || it->first != id) return false;
    *slot = it->second;
    return true;
  };
  auto agent_of = [&](uint32_t i, double pvx, double pvy) {
    cs_lp_agent a;
    a.agent_id = h.id[i];
    e->to_global(h.cell[i], h.off[i].x, h.off[i].y, &a.x, &a.y);
    a.vx = h.vel[i].x;
    a.vy = h.vel[i].y;
    a.preferred_vx = pvx;
    a.preferred_vy = pvy;
    a.eyesight_range = e->groups[meta_group(e->gdev, h.meta[i])].eyesight;
    a.next_waypoint = meta_waypoint(e->gdev, h.meta[i]);
    return a;
  };
  std::vector<float2> out(n, make_float2(0.f, 0.f));
  for (size_t lp = 0; lp < by_lp.size(); ++lp) {
    std::vector<uint32_t>& slots = by_lp[lp];
    if (slots.empty()) continue;
    std::sort(slots.begin(), slots.end(), [&](uint32_t a, uint32_t b) { return h.id[a] < h.id[b]; });
    const size_t m = slots.size();
    std::vector<cs_lp_agent> agents(m);
    std::vector<double> xy(2 * m), radius(m), recommended(2 * m), answer(2 * m, 0.0);
    for (size_t k = 0; k < m; ++k) {
      const uint32_t i = slots[k];
      agents[k] = agent_of(i, rec[i].x, rec[i].y);
      xy[2 * k] = agents[k].x;
      xy[2 * k + 1] = agents[k].y;
      radius[k] = agents[k].eyesight_range;
      recommended[2 * k] = rec[i].x;
      recommended[2 * k + 1] = rec[i].y;
    }
    std::vector<uint32_t> ids, cells, counts;
    std::vector<float> d2;
    size_t cap = 64;
    for (;;) {  // (a crowd denser than the buffer: once more with room for the largest answer)
      if (int rc = radius_query_batch(e, m, xy.data(), radius.data(), cap, &ids, &d2, &cells, &counts, true, true)) return rc;
      const uint32_t most = counts.empty() ? 0u : *std::max_element(counts.begin(), counts.end());
      if (most <= cap) break;
      cap = (size_t)most + 16;
    }
    std::vector<uint64_t> nb_begin(m + 1, 0);
    std::vector<cs_lp_agent> neighbours;
    for (size_t k = 0; k < m; ++k) {
      for (uint32_t q = 0; q < counts[k]; ++q) {
        const uint32_t id = ids[k * cap + q];
        uint32_t slot = 0;
        if (id == agents[k].agent_id || !find_slot(id, &slot)) continue;  // itself: lib.rs:284
        neighbours.push_back(agent_of(slot, 0.0, 0.0));
      }
      nb_begin[k + 1] = neighbours.size();
    }
    if (neighbours.empty()) neighbours.emplace_back();  // (a valid pointer for an empty list)
    if (e->lp_callbacks[lp].fn(e->lp_callbacks[lp].user, m, agents.data(), recommended.data(), nb_begin.data(),
                               neighbours.data(), answer.data()) != 0) {
      e->error = "a host LocalPlanner failed (its callback returned non-zero): nothing was committed";
      return 9;
    }
    for (size_t k = 0; k < m; ++k) out[slots[k]] = make_float2((float)answer[2 * k], (float)answer[2 * k + 1]);
  }
  if (hipMemcpyAsync(e->lp_vel, out.data(), (size_t)n * sizeof(float2), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
      hipStreamSynchronize(e->stream) != hipSuccess) {  // (`out` leaves scope)
    e->error = "HIP error in a host local planner's turn";
    return 90;
  }
  return 0;
}

// k nearest agents of n points at once: batches of radius queries with a doubling radius for the
// points that do not hold k agents yet, then the k smallest distances (ties by ascending id).
int cs_query_knn_batch(cs_engine* e, size_t n, const double* xy, size_t k, uint64_t* out_ids, uint64_t* out_counts,
                       float* out_d2) {
  hipSetDevice(e->device);
  for (size_t i = 0; i < n && out_counts; ++i) out_counts[i] = 0;
  if (n == 0 || k == 0) return 0;
  if (int rc = e->refresh_counts()) return rc;
  const uint64_t population = e->n_alive_host;
  std::vector<double> r(n, e->grid.cell_size), qxy, qr;
  std::vector<size_t> todo(n), next;
  for (size_t i = 0; i < n; ++i) todo[i] = i;
  size_t cap = std::max<size_t>(2 * k + 16, 32);
  while (!todo.empty()) {
    qxy.resize(2 * todo.size());
    qr.resize(todo.size());
    for (size_t t = 0; t < todo.size(); ++t) {
      qxy[2 * t] = xy[2 * todo[t]];
      qxy[2 * t + 1] = xy[2 * todo[t] + 1];
      qr[t] = r[todo[t]];
    }
    std::vector<uint32_t> ids, cells, counts;
    std::vector<float> d2;
    if (int rc = radius_query_batch(e, todo.size(), qxy.data(), qr.data(), cap, &ids, &d2, &cells, &counts, false)) return rc;
    next.clear();
    bool grow_cap = false;
    for (size_t t = 0; t < todo.size(); ++t) {
      const size_t i = todo[t];
      const double reach = 2.0 * (std::fabs(xy[2 * i] - e->grid.offset_x) + std::fabs(xy[2 * i + 1] - e->grid.offset_y) +
                                  e->grid.width + e->grid.height);
      if (counts[t] > cap) {  // more in sight than the buffer holds: same radius, larger buffer
        grow_cap = true;
        next.push_back(i);
        continue;
      }
      std::vector<std::pair<float, uint32_t>> by_dist(counts[t]);
      for (size_t m = 0; m < counts[t]; ++m) by_dist[m] = {d2[t * cap + m], ids[t * cap + m]};
      std::sort(by_dist.begin(), by_dist.end());
      by_dist.erase(std::unique(by_dist.begin(), by_dist.end()), by_dist.end());  // aliased cells list a member twice
      if (by_dist.size() < k && by_dist.size() < population && r[i] <= reach) {
        r[i] *= 2.0;
        next.push_back(i);
        continue;
      }
      const size_t m = std::min(k, by_dist.size());
      for (size_t j = 0; j < m; ++j) {
        out_ids[i * k + j] = by_dist[j].second;
        if (out_d2) out_d2[i * k + j] = by_dist[j].first;
      }
      if (out_counts) out_counts[i] = m;
    }
    if (grow_cap) cap *= 4;
    todo.swap(next);
  }
  return 0;
}

// SpatialIndex::get_neighbours_in_radius, location_hash_2d.rs:240-258
size_t cs_query_radius(cs_engine* e, double radius, double x, double y, uint64_t* out_ids, size_t cap) {
  hipSetDevice(e->device);
  std::vector<uint32_t> ids;
  std::vector<float> d2;
  size_t cnt = radius_query(e, radius, x, y, &ids, &d2, cap, true);
  for (size_t i = 0; i < ids.size() && i < cap; ++i) out_ids[i] = ids[i];
  return cnt;
}

// SpatialIndex::get_nearest_neighbours, location_hash_2d.rs:151-238.  Not on the step path
// (SURVEY.md §8a row a14).  The reference's ring search has defects (it never visits one corner
// of a ring, visits another twice and stops as soon as it holds n candidates); this is the exact
// k-NN instead: radius queries on the device index with a doubling radius until k agents are
// inside, then the k smallest distances (ties by ascending id).  Cost O(neighbourhood), not O(N).
size_t cs_query_knn(cs_engine* e, size_t k, double x, double y, uint64_t* out_ids) {
  hipSetDevice(e->device);
  if (k == 0 || e->refresh_counts() != 0 || e->n_alive_host == 0) return 0;
  const double reach = 2.0 * (std::fabs(x - e->grid.offset_x) + std::fabs(y - e->grid.offset_y) +
                              e->grid.width + e->grid.height);
  std::vector<uint32_t> ids;
  std::vector<float> d2;
  std::vector<std::pair<float, uint32_t>> by_dist;
  double r = e->grid.cell_size;
  for (;;) {
    radius_query(e, r, x, y, &ids, &d2, (size_t)1 << 22, false);
    // a window wider than the grid visits aliased cells twice (row stride nx on both axes,
    // location_hash_2d.rs:59), so count distinct ids
    by_dist.resize(ids.size());
    for (size_t i = 0; i < ids.size(); ++i) by_dist[i] = {d2[i], ids[i]};
    std::sort(by_dist.begin(), by_dist.end());
    by_dist.erase(std::unique(by_dist.begin(), by_dist.end()), by_dist.end());
    if (by_dist.size() >= k || by_dist.size() >= e->n_alive_host || r > reach) break;
    r *= 2.0;
  }
  size_t n = std::min(k, by_dist.size());
  for (size_t i = 0; i < n; ++i) out_ids[i] = by_dist[i].second;
  return n;
}

#ifdef CS_PHASE_CLOCKS
// profiling build only: read (and optionally clear) the per-phase wave cycles of the tiled kernel
void cs_debug_phase_cycles(cs_engine* e, unsigned long long* out, int reset) {
  hipSetDevice(e->device);
  hipStreamSynchronize(e->stream);
  static unsigned long long host[PHASE_SLOTS][PHASE_COUNT];
  hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_cycles), sizeof host);
  for (int k = 0; k < PHASE_COUNT; ++k) {
    out[k] = 0;
    for (int q = 0; q < PHASE_SLOTS; ++q) out[k] += host[q][k];
  }
  if (reset) {
    memset(host, 0, sizeof host);
    hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), host, sizeof host);
  }
}
#endif

void cs_profile_enable(cs_engine* e, uint32_t kernel_mask) { e->profiling = kernel_mask; }
void cs_profile_stride(cs_engine* e, uint32_t every) { e->prof_stride = every ? every : 1u; }
int cs_profile_read(cs_engine* e, uint32_t kernel, double* total_ms, uint64_t* launches) {
  if (kernel >= CS_K_COUNT) return 2;
  hipSetDevice(e->device);
  e->prof_collect();
  if (total_ms) *total_ms = e->prof_ms[kernel];
  if (launches) *launches = e->prof_n[kernel];
  return 0;
}
void cs_profile_reset(cs_engine* e) {
  hipSetDevice(e->device);
  e->prof_collect();
  for (int k = 0; k < CS_K_COUNT; ++k) {
    e->prof_ms[k] = 0;
    e->prof_n[k] = 0;
  }
}

int cs_halo_set_buffers(cs_engine* e, uint32_t dir, void* send_dev, void* recv_dev,
                        uint64_t capacity_records) {
  hipSetDevice(e->device);
  if (!e->tile || dir > 7 || capacity_records == 0 || capacity_records > 0x7FFFFFFFull) {
    e->error = "halo buffers need a tile engine, a direction 0..7 and a capacity";
    return 3;
  }
  e->halo[dir].send = static_cast<HaloRecord*>(send_dev);
  e->halo[dir].recv = static_cast<HaloRecord*>(recv_dev);
  e->halo[dir].cap = (uint32_t)capacity_records;
  if (dir < 4) e->halo_counts_clean[dir / 2] = false;
  e->halo_all_clean = false;
  e->halo_invalidate();
  // room for everything the four neighbours may deliver in one step
  return e->reserve((uint64_t)e->n_slots + 1024);
}

int cs_halo_pack_all(cs_engine* e) {
  hipSetDevice(e->device);
  return e->halo_pack_all();
}

int cs_halo_unpack_all(cs_engine* e) {
  hipSetDevice(e->device);
  return e->halo_unpack_all();
}

size_t cs_spawn_probe(cs_engine* e, double dt_seconds, uint8_t* flags, size_t cap) {
  hipSetDevice(e->device);
  if (!e->tile || cap < e->sinks.size()) {
    e->error = "cs_spawn_probe needs a tile engine and room for one flag per source-sink";
    return SIZE_MAX;
  }
  if (e->spawn_probe(dt_seconds, flags) != 0) return SIZE_MAX;
  return e->sinks.size();
}

int cs_spawn_commit(cs_engine* e, const uint8_t* flags, size_t n) {
  hipSetDevice(e->device);
  if (!e->tile || n != e->sinks.size()) {
    e->error = "cs_spawn_commit needs a tile engine and one flag per source-sink";
    return 3;
  }
  return e->spawn_commit(flags);
}

int cs_spawn_probe_dev(cs_engine* e, double dt_seconds, int* flags_dev, size_t cap) {
  hipSetDevice(e->device);
  if (!e->tile || cap < e->sinks.size() || !flags_dev) {
    e->error = "cs_spawn_probe_dev needs a tile engine and room for one flag per source-sink";
    return 3;
  }
  if (e->record_events || e->any_callback_hlp) {
    e->error = "cs_spawn_probe_dev: listeners and host planners need the host-side cs_spawn_probe";
    return 3;
  }
  return e->spawn_probe_dev(dt_seconds, flags_dev);
}

int cs_spawn_commit_dev(cs_engine* e, const int* flags_dev, size_t n) {
  hipSetDevice(e->device);
  if (!e->tile || n < e->sinks.size() || !flags_dev) {
    e->error = "cs_spawn_commit_dev needs a tile engine and one flag per source-sink";
    return 3;
  }
  return e->spawn_commit_dev(flags_dev);
}

int cs_halo_pack(cs_engine* e, uint32_t axis) {
  hipSetDevice(e->device);
  if (!e->tile || axis > 1) {
    e->error = "tiles are not enabled on this engine";
    return 3;
  }
  return e->halo_pack(axis);
}

int cs_halo_unpack(cs_engine* e, uint32_t axis) {
  hipSetDevice(e->device);
  if (!e->tile || axis > 1) {
    e->error = "tiles are not enabled on this engine";
    return 3;
  }
  return e->halo_unpack(axis);
}

// ---- re-cutting a running mesh ----
int cs_tile_histogram(cs_engine* e, uint64_t* rows, uint64_t* cols) {
  hipSetDevice(e->device);
  return e->tile_histogram(rows, cols);
}
size_t cs_tile_export(cs_engine* e, void* records, size_t cap_records) {
  hipSetDevice(e->device);
  const long long n = e->tile_export(static_cast<HaloRecord*>(records), cap_records);
  return n < 0 ? SIZE_MAX : (size_t)n;
}
int cs_tile_retile(cs_engine* e, uint32_t cx0, uint32_t cx1, uint32_t cy0, uint32_t cy1) {
  hipSetDevice(e->device);
  return e->tile_retile(cx0, cx1, cy0, cy1);
}
int cs_tile_import(cs_engine* e, const void* records, size_t n) {
  hipSetDevice(e->device);
  return e->tile_import(static_cast<const HaloRecord*>(records), n);
}

// ---- route followers on tiles: keeping every tile's route book alike ----
size_t cs_route_misses(cs_engine* e, cs_route_miss* out, size_t cap) {
  const size_t n = std::min(cap, e->route_misses.size());
  for (size_t k = 0; k < n && out; ++k) out[k] = e->route_misses[k];
  return e->route_misses.size();
}

int cs_route_resolve(cs_engine* e, const cs_route_miss* all, size_t n) {
  hipSetDevice(e->device);
  for (size_t k = 0; k < n; ++k) {
    const cs_route_miss& m = all[k];
    if (m.hlp >= e->hlps.size() || e->hlps[m.hlp].kind != CS_HLP_ROUTE) {
      e->error = "cs_route_resolve: not a route planner";
      return 2;
    }
    const uint32_t st = e->route_lookup(m.hlp, m.px, m.py, m.tx, m.ty);  // plans on a miss: the book grows
    if (!st) continue;  // "Failed to find contiguous path": the agent keeps what it had
    for (const cs_route_miss& mine : e->route_misses)
      if (mine.id == m.id) e->route_pending.push_back(make_uint2(mine.slot, st));
  }
  e->route_misses.clear();
  if (int rc = e->flush_route_tables()) return rc;
  return e->flush_route_pending();
}

// ---- the transport itself, over RCCL (cs_rccl.hip.inc) ----
int cs_rccl_unique_id(uint8_t* out_id) {
  rccl_api::Api& a = rccl_api::api();
  if (!a.handle || !out_id) return 8;
  rccl_api::UniqueId id;
  if (a.get_unique_id(&id) != 0) return 8;
  std::memcpy(out_id, id.internal, CS_RCCL_UNIQUE_ID_BYTES);
  return 0;
}

int cs_rccl_comm_init(cs_engine* e, int32_t n_ranks, int32_t rank, const uint8_t* id) {
  hipSetDevice(e->device);
  rccl_api::Api& a = rccl_api::api();
  if (!a.handle) {
    e->error = a.why;
    return 8;
  }
  if (!e->tile || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
    e->error = "cs_rccl_comm_init needs a tile engine, a unique id and 0 <= rank < n_ranks";
    return 3;
  }
  if (e->rccl_comm && e->rccl_comm_owned) a.comm_destroy(e->rccl_comm);
  e->rccl_comm = nullptr;
  rccl_api::UniqueId uid;
  std::memcpy(uid.internal, id, CS_RCCL_UNIQUE_ID_BYTES);
  rccl_api::Comm comm = nullptr;
  if (!rccl_api::ok(e, a.comm_init_rank(&comm, n_ranks, uid, rank), "ncclCommInitRank")) return 8;
  e->rccl_comm = comm;
  e->rccl_comm_owned = true;
  return 0;
}

int cs_rccl_comm_adopt(cs_engine* e, void* nccl_comm) {
  rccl_api::Api& a = rccl_api::api();
  if (!a.handle) {
    e->error = a.why;
    return 8;
  }
  if (e->rccl_comm && e->rccl_comm_owned) a.comm_destroy(e->rccl_comm);
  e->rccl_comm = nccl_comm;
  e->rccl_comm_owned = false;
  return 0;
}

int cs_halo_set_peers(cs_engine* e, const int32_t* peers8) {
  if (!e->tile || !peers8) {
    e->error = "cs_halo_set_peers needs a tile engine and eight ranks (-1 = no neighbour)";
    return 3;
  }
  for (int d = 0; d < 8; ++d) e->halo_peer[d] = peers8[d];
  return 0;
}

static int halo_exchange_rccl_on(cs_engine* e, int32_t axis, hipStream_t stream);

int cs_halo_exchange_rccl(cs_engine* e, int32_t axis) {
  hipSetDevice(e->device);
  if (e->exchanged_ahead && axis < 0) {
    // the previous cs_tile_step_rccl already exchanged this state's halo on its second stream,
    // behind the border windows' launch: wait for it (in stream order) instead of sending again
    e->exchanged_ahead = false;
    e->n_exchanges_ahead_used += 1;
    return hipStreamWaitEvent(e->stream, e->ev_xchg, 0) == hipSuccess ? 0 : 90;
  }
  if (e->aux_stream && e->ev_xchg)  // (an exchange ahead that a later change of the agents made void)
    if (hipStreamWaitEvent(e->stream, e->ev_xchg, 0) != hipSuccess) return 90;
  return halo_exchange_rccl_on(e, axis, e->stream);
}

static int halo_exchange_rccl_on(cs_engine* e, int32_t axis, hipStream_t stream) {
  rccl_api::Api& a = rccl_api::api();
  if (!a.handle || !e->rccl_comm) {
    e->error = a.handle ? "no RCCL communicator: call cs_rccl_comm_init or cs_rccl_comm_adopt first" : a.why;
    return 8;
  }
  const int d0 = axis < 0 ? 0 : 2 * axis, d1 = axis < 0 ? 8 : 2 * axis + 2;
  if (axis > 1) {
    e->error = "cs_halo_exchange_rccl: axis is 0, 1 or negative (all eight directions)";
    return 3;
  }
  const int timed = e->prof_open_on(CS_K_HALO_EXCHANGE, stream);
  bool good = rccl_api::ok(e, a.group_start(), "ncclGroupStart");
  for (int d = d0; good && d < d1; ++d) {
    const cs_engine::HaloDir& h = e->halo[d];
    if (!h.send || !h.recv || e->halo_peer[d] < 0) continue;
    const size_t bytes = ((size_t)h.cap + 1u) * sizeof(HaloRecord);
    good = rccl_api::ok(e, a.send(h.send, bytes, rccl_api::kUint8, e->halo_peer[d], e->rccl_comm, stream), "ncclSend") &&
           rccl_api::ok(e, a.recv(h.recv, bytes, rccl_api::kUint8, e->halo_peer[d], e->rccl_comm, stream), "ncclRecv");
  }
  const int end_rc = a.group_end();  // always close the group
  e->prof_close_on(timed, stream);
  if (!good) return 8;
  return rccl_api::ok(e, end_rc, "ncclGroupEnd") ? 0 : 8;
}

int cs_allreduce_max_i32_rccl(cs_engine* e, int* values_dev, size_t n) {
  hipSetDevice(e->device);
  rccl_api::Api& a = rccl_api::api();
  if (!a.handle || !e->rccl_comm) {
    e->error = a.handle ? "no RCCL communicator: call cs_rccl_comm_init or cs_rccl_comm_adopt first" : a.why;
    return 8;
  }
  if (n == 0) return 0;
  // Every operation on the communicator is ordered against every other ON EVERY RANK ALIKE: an exchange made ahead on the
  // second stream (CS_CFG_TILE_OVERLAP) comes first, then this (the engine's stream waits for its event; a no-op when
  // none was made).  Two operations of one communicator in flight on two streams in an order that differs between
  // ranks is a deadlock RCCL does not report.
  if (e->aux_stream && e->ev_xchg && hipStreamWaitEvent(e->stream, e->ev_xchg, 0) != hipSuccess) return 90;
  return rccl_api::ok(e, a.all_reduce(values_dev, values_dev, n, rccl_api::kInt32, rccl_api::kMax, e->rccl_comm, e->stream),
                      "ncclAllReduce") ? 0 : 8;
}

// The whole multi-GPU step of a tile in ONE call, on the engine's stream, nothing waiting for the
// host: halo pack -> RCCL exchange -> unpack -> (source-sinks: device-side probe, all-reduce of the
// flags, commit) -> cs_step.  For hosts without listeners, host planners or multi-leg route sinks
// (those need the split calls: events must reach the host between the phases).
// e->step_colls tells which of the step's collectives this call has issued (bit 0: the halo exchange, bit 1: the
// all-reduce of the spawn flags): a mesh whose tile fails half way issues the missing ones itself, so that the other
// ranks' collectives still find their partner (cs_mesh.hip.inc: mesh_zombie_step).
int cs_tile_step_rccl(cs_engine* e, double dt_seconds, cs_step_report* report) {
  hipSetDevice(e->device);
  e->step_colls = 0;
  if (!e->tile) {
    e->error = "cs_tile_step_rccl needs a tile engine";
    return 3;
  }
  if (int rc = e->halo_pack_all()) return rc;
  if (int rc = cs_halo_exchange_rccl(e, -1)) return rc;
  e->step_colls |= 1u;
  if (int rc = e->halo_unpack_all()) return rc;
  if (int rc = e->upload_sinks()) return rc;
  if (int rc = e->upload_groups()) return rc;
  // (gated on the sink SLOTS, which every rank counts alike and which never shrink: a rank that takes the split
  // calls instead all-reduces under the same condition, DistributedTiles.step)
  if (!e->sinks.empty()) {
    if (e->n_live_sinks > 0 && (e->record_events || e->any_callback_hlp)) {
      e->error = "cs_tile_step_rccl: listeners and host planners need the split calls (cs_spawn_probe / cs_spawn_commit)";
      return 3;
    }
    const size_t ns = e->sinks.size();
    if (int rc = e->reserve_step_flags(ns)) return rc;
    if (int rc = e->spawn_probe_dev(dt_seconds, e->step_flags_dev)) return rc;
    if (int rc = cs_allreduce_max_i32_rccl(e, e->step_flags_dev, ns)) return rc;
    e->step_colls |= 2u;
    if (int rc = e->spawn_commit_dev(e->step_flags_dev)) return rc;
  }
  // CS_CFG_TILE_OVERLAP: the step puts the border windows' launch on the second stream; the next
  // step's exchange follows them there while the interior windows still run on the engine's stream
  e->overlap_ready = e->split_margin() != 0u && e->rccl_comm != nullptr && !e->no_exchange_ahead;
  const int rc = e->step(dt_seconds, report);
  e->overlap_ready = false;
  if (rc == 0) e->steps_done += 1;
  if (rc == 0 && e->border_on_aux && e->halo_prepacked) {
    e->border_on_aux = false;
    if (e->early_pending) {  // (the border windows are the first workgroups of the ONE launch: wait for the last of them)
      e->early_pending = false;
      hipLaunchKernelGGL(k_wait_border, dim3(1), dim3(64), 0, e->aux_stream, e->ctr, e->early_seq);
    }
    if (int rc2 = halo_exchange_rccl_on(e, -1, e->aux_stream)) return rc2;
    HIP_OK_E(e, hipEventRecord(e->ev_xchg, e->aux_stream));
    e->exchanged_ahead = true;
    e->n_exchanges_ahead += 1;
  }
  return rc;
}

// The collectives of one cs_tile_step_rccl and nothing else: what a tile that has FAILED issues in the place of its
// steps until the ranks of its mesh have agreed that the run is over (the others' ncclSend / ncclRecv / ncclAllReduce
// would wait for it for ever: RCCL has no timeout).  `done`: the collectives a failed cs_tile_step_rccl call issued
// before it gave up (cs_engine::step_colls), 0 for a whole step.  The records that travel are whatever the send
// buffers hold; nobody steps on them.
static int tile_zombie_collectives(cs_engine* e, uint32_t done) {
  hipSetDevice(e->device);
  if (!(done & 1u)) {
    if (e->exchanged_ahead) {  // (made by the last good step, on the second stream: it IS this step's exchange)
      e->exchanged_ahead = false;
      if (hipStreamWaitEvent(e->stream, e->ev_xchg, 0) != hipSuccess) return 90;
    } else if (int rc = halo_exchange_rccl_on(e, -1, e->stream)) {
      return rc;
    }
  }
  if (!(done & 2u) && !e->sinks.empty()) {
    const size_t ns = e->sinks.size();
    if (int rc = e->reserve_step_flags(ns)) return rc;
    if (int rc = cs_allreduce_max_i32_rccl(e, e->step_flags_dev, ns)) return rc;
  }
  return 0;
}

}  // extern "C"

#include "cs_mesh.hip.inc"
