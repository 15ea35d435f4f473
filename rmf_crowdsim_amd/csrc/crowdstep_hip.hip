// crowdstep_hip.hip — MI355X (gfx950) engine behind include/crowdstep.h.
//
// The per-step hot path of rmf_crowdsim's Simulation::step (reference
// rmf_crowdsim/src/lib.rs:195-383) as HIP kernels over cell-sorted SoA agent
// state.  See DESIGN.md for the data layout and the per-kernel rooflines.
//
//   k_count     cell histogram + arrival rank            (location_hash_2d.rs:126-149)
//   k_scan_*    exclusive scan of the cell counts        (implicit in Vec<HashSet>, :15)
//   k_scatter   reorder records into cell order          (:139-147)
//   k_step_*    neighbour query + Zanlungo + integrate + re-bin + waypoint/sink test
//               (location_hash_2d.rs:240-258, zanlungo.rs:49-217, lib.rs:259-359)
//   k_spawn     source occupancy + append                (lib.rs:199-254)
//
// Device state is f32 and CELL-RELATIVE: an agent is (stored cell, offset from
// that cell's origin), so relative positions between neighbours keep ~1e-7 m
// resolution at any domain size.  f64 appears only at the ABI.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "crowdstep.h"

#define CS_INVALID_CELL 0xFFFFFFFFu
#define CS_MAX_GROUPS 65535u  // the group index travels in 16 bits of `meta`
#define CS_SPAWN_OCCUPANCY_RADIUS 0.4  // hard-coded in the reference, lib.rs:212-214

// ---------------------------------------------------------------------------
// device-side tables
// ---------------------------------------------------------------------------
struct GridDev {
  uint32_t nx;      // row stride AND number of x rows: (width / cell) as usize
  uint32_t ny;      // (height / cell) as usize
  uint32_t ncells;  // nx * ny
  float cs;         // cell size, f32
  float cs_lo;      // cell_size - (double)cs, second word for exact re-basing
  float inv_cs;
  // tile mode (one engine per GPU): the local grid is the owned cell rectangle
  // [own_x0, own_x1) x [own_y0, own_y1) (local cell coordinates) plus its ghost ring; local
  // cell (0,0) is global cell (org_x, org_y).  Not a tile: tile = 0 and the owned rectangle
  // is the whole grid.
  uint32_t tile;
  uint32_t own_x0, own_x1, own_y0, own_y1;
  uint32_t org_x, org_y;
};

// One group per add_agents call / per source-sink: the reference passes the
// planners and the eyesight per call (lib.rs:119-125, source_sink.rs:46-59).
struct GroupDev {
  float eyesight;
  uint32_t lp_kind;  // 0 = NoLocalPlan, 1 = Zanlungo
  float A, D, inv_mass, R;
  uint32_t hlp_kind;  // CS_HLP_*
  float hvx, hvy;
  int32_t sink;  // owning source-sink slot or -1
};

struct SinkDev {
  double src_x, src_y;  // global, for events
  uint32_t src_cell;    // stored cell of the source point (or CS_INVALID_CELL)
  float src_ox, src_oy;
  float radius_sink;
  uint32_t wp_begin, wp_count;  // into the waypoint array (global f64 pairs)
  uint32_t loop_forever;
  uint32_t group;
  float eyesight;
};

struct Counters {
  // persistent
  uint32_t n_alive;          // written by the scan: total of the histogram
  uint32_t n_out_of_bounds;  // cumulative: any non-zero value poisons the engine
  uint32_t n_owned;          // agents in the owned rectangle (tile mode; else = n_alive)
  uint32_t n_pending;        // slots in use in the unsorted buffer (halo unpack appends)
  // written before the re-sort of a step (zeroed by the host when sinks / tiles exist)
  uint32_t n_spawned;
  uint32_t n_halo_overflow;
  // written by the step kernel (zeroed by the scan of the same step)
  uint32_t n_destroyed;
  uint32_t n_waypoint_hits;
  uint32_t n_tti_zero;
  uint32_t n_nonfinite;
  uint32_t n_clamped;
  uint32_t n_wp_events;
  uint32_t pad[4];
};

struct AgentArrays {
  float2* off;
  float2* vel;
  uint32_t* id;
  uint32_t* cell;  // stored flat cell (reference's location_to_index), or CS_INVALID_CELL
  uint32_t* meta;  // group (low 16) | next_waypoint (high 16)
  uint32_t* rank;  // arrival rank inside `cell` for the next scatter
};

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float f_inf() { return __builtin_huge_valf(); }
__device__ __forceinline__ float f_nan() { return __builtin_nanf(""); }

// ---------------------------------------------------------------------------
// Arithmetic of the neighbour pass.  One definition, used by both neighbour
// kernels, so their results are bitwise equal.  Divisions, square roots and the
// exponential use the 1-ulp hardware forms (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 /
// v_exp_f32): the IEEE-exact expansions cost ~10 VALU each and this kernel is
// VALU-issue bound (profiles/), while the fp32 tolerance of the path is 1e-4.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Zanlungo::time_to_collision, zanlungo.rs:49-74 (collision distance is R, not 2R).
// With bh = b/2 the quadratic reads t = (-bh -+ sqrt(bh^2 - a c)) / a: the same values as
// the reference's (-b -+ sqrt(b^2 - 4ac)) / 2a, the factors of two being exact.  For a > 0
// the roots are ordered (t0 <= t1), which folds the reference's case analysis into
//   disc < 0 -> inf;  t0 > 0 -> t0;  t1 > 0 -> (t0 < 0 ? 0 : t1);  else inf.
// f32 has 1/8 of f64's exponent range: |rel_vel|^2 underflows near 1e-19 m/s where the
// reference's f64 does not, and a flushed `a` with b != 0 would read as "colliding now"
// (t0 = -inf, t1 = +inf -> 0).  Tiny relative velocities are therefore scaled by 2^48
// first (t(s*rv) = t(rv)/s exactly); what still underflows takes the a -> 0 limit of
// the same quadratic, which is what f64 computes there.  d2 = |rp|^2.
// Rare path: |rel_vel| < 1e-12 m/s.
__device__ __noinline__ float ttc_tiny_f32(float rvx, float rvy, float rpx, float rpy, float d2,
                                           float R2) {
  const float tscale = 0x1p48f;
  rvx *= tscale;
  rvy *= tscale;
  const float a = __builtin_fmaf(rvx, rvx, rvy * rvy);
  const float bh = __builtin_fmaf(rvx, rpx, rvy * rpy);
  const float c = d2 - R2;
  if (a < 1e-30f) {
    if (bh == 0.0f) return f_inf();  // equal velocities: 0/0 = NaN fails every comparison
    if (c < 0.0f) return 0.0f;       // already inside R: t0 < 0 < t1
    return bh < 0.0f ? (-c / (2.0f * bh)) * tscale : f_inf();
  }
  const float disc = __builtin_fmaf(bh, bh, -(a * c));
  const float root = fast_sqrt(disc);
  const float ia = fast_rcp(a) * tscale;
  const float t0 = (-bh - root) * ia;
  const float t1 = (-bh + root) * ia;
  float t = (t1 > 0.0f) ? ((t0 < 0.0f) ? 0.0f : t1) : f_inf();
  t = (t0 > 0.0f) ? t0 : t;
  return (disc < 0.0f) ? f_inf() : t;
}

__device__ __forceinline__ float ttc_f32(float rvx, float rvy, float rpx, float rpy, float d2,
                                         float R2) {
  const float rvmax = fmaxf(fabsf(rvx), fabsf(rvy));
  if (__builtin_expect(rvmax < 1e-12f, 0)) {
    // equal velocities (a lane of walkers): a = b = 0, 0/0 = NaN fails every comparison -> inf
    if (rvmax == 0.0f) return f_inf();
    return ttc_tiny_f32(rvx, rvy, rpx, rpy, d2, R2);
  }
  const float a = __builtin_fmaf(rvx, rvx, rvy * rvy);  // >= 1e-24 here
  const float bh = __builtin_fmaf(rvx, rpx, rvy * rpy);
  const float c = d2 - R2;
  const float disc = __builtin_fmaf(bh, bh, -(a * c));
  const float root = fast_sqrt(disc);
  const float ia = fast_rcp(a);
  const float t0 = (-bh - root) * ia;
  const float t1 = (-bh + root) * ia;
  float t = (t1 > 0.0f) ? ((t0 < 0.0f) ? 0.0f : t1) : f_inf();
  t = (t0 > 0.0f) ? t0 : t;
  return (disc < 0.0f) ? f_inf() : t;
}

// What every force term of one agent shares (zanlungo.rs:93-170 with weight = 2):
//   fut = v_i * t_i,  mag = min(1e15, 2 * A * |v_i| / t_i),  k = log2(e) / D
struct ForceCtx {
  float vix, viy, futx, futy, mag, two_R, k_exp;
};

__device__ __forceinline__ ForceCtx make_force_ctx(float vix, float viy, float T, const GroupDev& grp) {
  ForceCtx c;
  c.vix = vix;
  c.viy = viy;
  c.futx = vix * T;
  c.futy = viy * T;
  // weight * agent_scale * |my_vel - other_vel| / t_i with weight 2, other_vel 0; once per agent,
  // so the exact division is kept (t_i == 0 -> +inf -> clamped, zanlungo.rs:165-167)
  float mag = 2.0f * grp.A * sqrtf(__builtin_fmaf(vix, vix, viy * viy)) / T;
  c.mag = (mag >= 1e15f) ? 1e15f : mag;
  c.two_R = grp.R * 2.0f;
  c.k_exp = 1.44269504088896341f / grp.D;
  return c;
}

// Force on agent i from a neighbour j with the LARGER id (weight = 2 branch of
// compute_agent_force, zanlungo.rs:93-170, with right_of_way_vel :173-198 and slerp :23-28
// folded for the state the reference actually produces: a neighbour's preferred_vel is
// always (0,0) (lib.rs:140,261,271: it is set on the per-iteration clone only), so
// other_vel = v_j + 1*(0 - v_j) = 0 and the "stationary" branch :119-125 is the live one;
// slerp(1, d, perp, s) = d*(sin(0)/s) + perp*(sin(asin s)/s) = d*0 + perp*1 for s > 0 and NaN
// for s == 0, and the normalize() that follows removes the factor sin(asin s)/s = 1 +- 1 ulp).
// |perp| = |rp|, so the unit vector is perp * rsqrt(d2).   rp = p_j - p_i, d2 = |rp|^2.
__device__ __forceinline__ void zanlungo_forward_force(float rpx, float rpy, float d2,
                                                       const ForceCtx& c, float& fx, float& fy) {
  const float dx = c.futx - rpx, dy = c.futy - rpy;  // (p_i + v_i T) - (p_j + 0 T)
  const float dist = fast_sqrt(__builtin_fmaf(dx, dx, dy * dy));
  float px = rpy, py = -rpx;  // perp of q = p_i - p_j = -rp: (-q.y, q.x)
  const bool flip = __builtin_fmaf(px, c.vix, py * c.viy) < 0.0f;
  px = flip ? -px : px;
  py = flip ? -py : py;
  const float s = fabsf(__builtin_fmaf(px, dy, -(py * dx)));
  // s > 1 clamps to 1 and drops out; s == 0 or NaN poisons the direction (0/0)
  const float inv_n = (s > 0.0f) ? fast_rsq(d2) : f_nan();
  const float scale = c.mag * fast_exp2((c.two_R - dist) * c.k_exp) * inv_n;
  fx = __builtin_fmaf(px, scale, fx);
  fy = __builtin_fmaf(py, scale, fy);
}

struct StepParams {
  GridDev g;
  float dt;
  uint32_t n;  // slots in the sorted arrays (upper bound of alive)
  uint32_t has_sinks;
  uint32_t n_src_cells;  // 0 when no sources are binned
};

// Re-bin after integration: the reference's location_to_index
// (location_hash_2d.rs:54-66) on a cell-relative position.  Negative
// coordinates saturate to row/column 0, y beyond the stride aliases into the
// next row, flat >= len is "Index out of bounds".
__device__ __forceinline__ uint32_t rebin(const GridDev& g, uint32_t gx, uint32_t gy, float& ox,
                                          float& oy, Counters* ctr) {
  float kxf = floorf(ox * g.inv_cs), kyf = floorf(oy * g.inv_cs);
  if (kxf == 0.0f && kyf == 0.0f) return gx * g.nx + gy;  // fast path: same cell
  bool nanx = !(ox == ox), nany = !(oy == oy);
  // NaN as usize = 0 (saturating cast): binned into row/column 0, offset stays NaN
  long long cx = nanx ? 0 : (long long)gx + (long long)fminf(fmaxf(kxf, -4e9f), 4e9f);
  long long cy = nany ? 0 : (long long)gy + (long long)fminf(fmaxf(kyf, -4e9f), 4e9f);
  bool clamped = false;
  if (cx < 0) {
    cx = 0;
    clamped = !nanx;
  }
  if (cy < 0) {
    cy = 0;
    clamped = clamped || !nany;
  }
  if (clamped) atomicAdd(&ctr->n_clamped, 1u);
  unsigned long long flat = (unsigned long long)cx * g.nx + (unsigned long long)cy;
  // a tile has real neighbours instead of the reference's clamp / alias behaviour at its
  // edges: anything that leaves the local rectangle (more than a ghost ring in one step,
  // or out of the global grid) is an error
  if (flat >= g.ncells || (g.tile && (clamped || cy >= (long long)g.nx))) {
    atomicAdd(&ctr->n_out_of_bounds, 1u);
    return CS_INVALID_CELL;
  }
  // geometric coordinates of the STORED cell (differ from (cx,cy) when aliased)
  uint32_t sx = (uint32_t)cx, sy = (uint32_t)cy;
  if (cy >= g.nx) {
    sx = (uint32_t)(flat / g.nx);
    sy = (uint32_t)(flat % g.nx);
  }
  float mx = (float)((long long)sx - (long long)gx), my = (float)((long long)sy - (long long)gy);
  ox = (ox - mx * g.cs) - mx * g.cs_lo;
  oy = (oy - my * g.cs) - my * g.cs_lo;
  return (uint32_t)flat;
}

// Source occupancy for the next spawn phase: the reference asks the index for
// agents within 0.4 of each source (lib.rs:212-217); here every agent marks the
// sources it blocks.  Sources are binned once on the host (src_cell_start /
// src_sorted, same cell order as the agents).
__device__ __forceinline__ void mark_sources(const GridDev& g, const SinkDev* __restrict__ sinks,
                                             const uint32_t* __restrict__ src_cell_start,
                                             const uint32_t* __restrict__ src_sorted,
                                             uint32_t* __restrict__ src_occupied, uint32_t cell,
                                             float ox, float oy) {
  const float r = (float)CS_SPAWN_OCCUPANCY_RADIUS;
  if (!(fabsf(ox) < 1e6f * g.cs && fabsf(oy) < 1e6f * g.cs)) return;  // NaN / far outside
  const uint32_t sx = cell / g.nx, sy = cell - sx * g.nx;
  long long lx = (long long)sx + (long long)floorf((ox - r) * g.inv_cs - 1e-4f);
  long long hx = (long long)sx + (long long)floorf((ox + r) * g.inv_cs + 1e-4f);
  long long ly = (long long)sy + (long long)floorf((oy - r) * g.inv_cs - 1e-4f);
  long long hy = (long long)sy + (long long)floorf((oy + r) * g.inv_cs + 1e-4f);
  lx = max(lx, 0ll); ly = max(ly, 0ll);
  hx = min(hx, (long long)g.nx - 1); hy = min(hy, (long long)g.nx - 1);
  for (long long cx = lx; cx <= hx; ++cx) {
    for (long long cy = ly; cy <= hy; ++cy) {
      unsigned long long flat = (unsigned long long)cx * g.nx + (unsigned long long)cy;
      if (flat >= g.ncells) continue;
      uint32_t b = src_cell_start[flat], e = src_cell_start[flat + 1];
      for (uint32_t k = b; k < e; ++k) {
        uint32_t slot = src_sorted[k];
        float qx = (float)(cx - (long long)sx) * g.cs + (sinks[slot].src_ox - ox);
        float qy = (float)(cy - (long long)sy) * g.cs + (sinks[slot].src_oy - oy);
        if (sqrtf(qx * qx + qy * qy) < r) src_occupied[slot] = 1u;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K1: histogram + arrival rank (used when the histogram kept by k_step is stale)
// ---------------------------------------------------------------------------
__global__ void k_count(AgentArrays a, uint32_t first, uint32_t n, uint32_t* __restrict__ cell_count,
                        const Counters* __restrict__ ctr, uint32_t tile) {
  uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (tile && i >= ctr->n_pending)) return;
  uint32_t c = a.cell[i];
  if (c == CS_INVALID_CELL) return;
  a.rank[i] = atomicAdd(&cell_count[c], 1u);
}

__global__ void k_mark_sources(GridDev g, AgentArrays a, uint32_t n, const SinkDev* __restrict__ sinks,
                               const uint32_t* __restrict__ src_cell_start,
                               const uint32_t* __restrict__ src_sorted,
                               uint32_t* __restrict__ src_occupied, const Counters* __restrict__ ctr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (g.tile && i >= ctr->n_pending)) return;
  uint32_t c = a.cell[i];
  if (c == CS_INVALID_CELL) return;
  float2 o = a.off[i];
  mark_sources(g, sinks, src_cell_start, src_sorted, src_occupied, c, o.x, o.y);
}

// Tile engines: append agents whose ids were assigned across tiles (cs_spawn_commit).
struct SpawnRecord {
  float ox, oy;
  uint32_t id, cell, meta, pad;
};
__global__ void k_append_spawns(AgentArrays a, uint32_t slot_cap, const SpawnRecord* __restrict__ rec,
                                uint32_t n, uint32_t* __restrict__ cell_count, Counters* __restrict__ ctr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SpawnRecord r = rec[i];
  const uint32_t slot = atomicAdd(&ctr->n_pending, 1u);
  if (slot >= slot_cap) {
    atomicAdd(&ctr->n_halo_overflow, 1u);
    return;
  }
  a.off[slot] = make_float2(r.ox, r.oy);
  a.vel[slot] = make_float2(0.0f, 0.0f);
  a.id[slot] = r.id;
  a.cell[slot] = r.cell;
  a.meta[slot] = r.meta;
  a.rank[slot] = atomicAdd(&cell_count[r.cell], 1u);
}

// ---------------------------------------------------------------------------
// K2: exclusive scan of cell_count -> cell_start (and zero cell_count)
//   pass A: per-block totals; pass B: block offset by summing earlier totals,
//   then an in-block scan.  1024 cells per block.
// ---------------------------------------------------------------------------
#define SCAN_BLOCK 256
#define SCAN_ITEMS 4
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_totals(const uint32_t* __restrict__ cell_count,
                                                            uint32_t ncells,
                                                            uint32_t* __restrict__ block_totals) {
  __shared__ uint32_t wsum[SCAN_BLOCK / 64];
  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  if (base + SCAN_ITEMS <= ncells) {
    uint4 v = *reinterpret_cast<const uint4*>(cell_count + base);
    s = v.x + v.y + v.z + v.w;
  } else {
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
      if (base + k < ncells) s += cell_count[base + k];
  }
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += wsum[w];
    block_totals[blockIdx.x] = t;
  }
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(uint32_t* __restrict__ cell_count,
                                                           uint32_t ncells,
                                                           const uint32_t* __restrict__ block_totals,
                                                           uint32_t nblocks,
                                                           uint32_t* __restrict__ cell_start,
                                                           Counters* __restrict__ ctr,
                                                           uint32_t* __restrict__ n_blocks) {
  __shared__ uint32_t wsum[SCAN_BLOCK / 64];
  __shared__ uint32_t s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // offset of this block = sum of the totals of the blocks before it
  uint32_t part = 0;
  for (uint32_t b = threadIdx.x; b < blockIdx.x; b += SCAN_BLOCK) part += block_totals[b];
  for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
  if (lane == 0) wsum[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) t += wsum[w];
    s_base = t;
  }
  __syncthreads();
  const uint32_t block_base = s_base;
  __syncthreads();

  uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  bool full = base + SCAN_ITEMS <= ncells;
  if (full) {
    uint4 q = *reinterpret_cast<const uint4*>(cell_count + base);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    *reinterpret_cast<uint4*>(cell_count + base) = make_uint4(0, 0, 0, 0);
  } else {
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) {
      v[k] = (base + k < ncells) ? cell_count[base + k] : 0u;
      if (base + k < ncells) cell_count[base + k] = 0u;
    }
  }
  uint32_t tsum = v[0] + v[1] + v[2] + v[3];
  uint32_t incl = wave_incl_scan(tsum, lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  uint32_t excl = block_base + woff + incl - tsum;
  uint32_t o0 = excl, o1 = o0 + v[0], o2 = o1 + v[1], o3 = o2 + v[2];
  if (full) {
    *reinterpret_cast<uint4*>(cell_start + base) = make_uint4(o0, o1, o2, o3);
  } else {
    uint32_t o[4] = {o0, o1, o2, o3};
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
      if (base + k < ncells) cell_start[base + k] = o[k];
  }
  // the thread holding the last cell publishes the grand total
  if (base <= ncells - 1 && ncells - 1 < base + SCAN_ITEMS) {
    uint32_t total = o3 + v[3];
    if (!full) {
      total = excl;
      for (uint32_t k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < ncells) total += v[k];
    }
    cell_start[ncells] = total;
    ctr->n_alive = total;
    ctr->n_owned = 0;  // the block builder that follows counts the owned agents
    *n_blocks = 0;     // ... and the workgroups of the step kernel
    ctr->n_destroyed = 0;  // the step kernel that follows counts into these
    ctr->n_waypoint_hits = 0;
    ctr->n_tti_zero = 0;
    ctr->n_nonfinite = 0;
    ctr->n_clamped = 0;
    ctr->n_wp_events = 0;
  }
}

// ---------------------------------------------------------------------------
// K3: scatter records into cell order
// ---------------------------------------------------------------------------
__global__ void k_scatter(AgentArrays src, AgentArrays dst, uint32_t n,
                          const uint32_t* __restrict__ cell_start, const Counters* __restrict__ ctr,
                          uint32_t tile) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || (tile && i >= ctr->n_pending)) return;
  uint32_t c = src.cell[i];
  if (c == CS_INVALID_CELL) return;
  uint32_t d = cell_start[c] + src.rank[i];
  dst.off[d] = src.off[i];
  dst.vel[d] = src.vel[i];
  dst.id[d] = src.id[i];
  dst.cell[d] = c;
  dst.meta[d] = src.meta[i];
}

// One returning atomic per distinct cell per wave instead of one per agent: the lanes that
// share a cell elect a leader, which reserves popcount(lanes) slots; rank = base + lane order.
// Groups are found first (ALU only), then every leader issues its atomic at once, so a wave
// pays one atomic latency, not one per distinct cell.
__device__ __forceinline__ uint32_t wave_histogram_rank(uint32_t* __restrict__ cell_count,
                                                        uint32_t cell, bool valid) {
  const int lane = __lane_id();
  int leader = lane;
  unsigned long long group = 0;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    const uint32_t c = __shfl(cell, first, 64);
    const bool same = valid && cell == c;
    const unsigned long long m = __ballot(same);
    if (same) {
      leader = first;
      group = m;
    }
    todo &= ~m;
  }
  uint32_t base = 0;
  if (valid && lane == leader) base = atomicAdd(&cell_count[cell], (uint32_t)__popcll(group));
  base = __shfl(base, leader, 64);
  return base + (uint32_t)__popcll(group & ((1ull << lane) - 1ull));
}

// ---------------------------------------------------------------------------
// Shared epilogue of the step kernels: integrate, re-bin, waypoint/sink test,
// histogram for the next scatter, source-occupancy marks.
// (lib.rs:295-346; commit lib.rs:350-359 is the write into the `out` arrays)
// ---------------------------------------------------------------------------
// set_target(&self.agents[&agent_id], waypoints[next], (r, r)): the agent as it was BEFORE the step
struct WpEvent {
  uint32_t id, group, next_wp, cell;
  float ox, oy;
};

struct EpilogueCtx {
  AgentArrays out;
  uint32_t* cell_count;
  Counters* ctr;
  const GroupDev* groups;
  const SinkDev* sinks;
  const double* waypoints;  // global f64 pairs
  double grid_off_x, grid_off_y, cell_size;
  uint2* destroyed;  // append list of (id, meta)
  uint32_t destroyed_cap;
  WpEvent* wp_events;  // waypoint advances of agents with a host planner (set_target, lib.rs:325-333)
  uint32_t wp_events_cap;
  // source occupancy for the next step's spawn phase
  const uint32_t* src_cell_start;  // per cell, into src_sorted
  const uint32_t* src_sorted;      // sink slots sorted by source cell
  uint32_t* src_occupied;
};

__device__ __forceinline__ void step_epilogue(const StepParams& P, const EpilogueCtx& E, uint32_t i,
                                              uint32_t gx, uint32_t gy, float2 off, uint32_t id,
                                              uint32_t meta, const GroupDev& grp, float wx, float wy) {
  // waypoint / sink test on the OLD position (lib.rs:304-336)
  uint32_t next_wp = meta >> 16;
  bool destroyed = false;
  if (grp.sink >= 0) {
    const SinkDev& s = E.sinks[grp.sink];
    if (next_wp >= s.wp_count) {
      destroyed = true;  // "rogue agent", lib.rs:310-313
    } else {
      double wxg = E.waypoints[2 * (s.wp_begin + next_wp)];
      double wyg = E.waypoints[2 * (s.wp_begin + next_wp) + 1];
      // waypoint relative to this agent's cell origin, rounded once to f32
      float rx = (float)(wxg - (E.grid_off_x + (double)(P.g.org_x + gx) * E.cell_size));
      float ry = (float)(wyg - (E.grid_off_y + (double)(P.g.org_y + gy) * E.cell_size));
      float ddx = off.x - rx, ddy = off.y - ry;
      if (sqrtf(ddx * ddx + ddy * ddy) < s.radius_sink) {
        atomicAdd(&E.ctr->n_waypoint_hits, 1u);
        if (next_wp == s.wp_count - 1) {
          if (s.loop_forever)
            next_wp = 0;
          else
            destroyed = true;
        } else {
          next_wp += 1;
          if (grp.hlp_kind == CS_HLP_CALLBACK) {
            uint32_t k = atomicAdd(&E.ctr->n_wp_events, 1u);
            if (k < E.wp_events_cap) E.wp_events[k] = WpEvent{id, meta & 0xFFFFu, next_wp, gx * P.g.nx + gy, off.x, off.y};
          }
        }
      }
    }
  }

  // integrate (lib.rs:295-297): new_pos = pos + vel * dt, cell-relative
  float nox = off.x + wx * P.dt, noy = off.y + wy * P.dt;
  if (!(fabsf(nox) < f_inf() && fabsf(noy) < f_inf() && fabsf(wx) < f_inf() && fabsf(wy) < f_inf()))
    atomicAdd(&E.ctr->n_nonfinite, 1u);
  uint32_t ncell = rebin(P.g, gx, gy, nox, noy, E.ctr);

  if (destroyed) {
    uint32_t k = atomicAdd(&E.ctr->n_destroyed, 1u);
    if (k < E.destroyed_cap) E.destroyed[k] = make_uint2(id, meta);
    ncell = CS_INVALID_CELL;
  }
  const bool keep = ncell != CS_INVALID_CELL;
  if (keep) {
    E.out.off[i] = make_float2(nox, noy);
    E.out.vel[i] = make_float2(wx, wy);
    E.out.id[i] = id;
    E.out.meta[i] = (meta & 0xFFFFu) | (next_wp << 16);
  }
  E.out.cell[i] = ncell;
  // histogram for the next scatter: agents are in cell order and few change cell per step,
  // so the lanes of a wave hit a handful of counters; one atomic per distinct cell per wave
  const uint32_t rank = wave_histogram_rank(E.cell_count, ncell, keep);
  if (!keep) return;
  E.out.rank[i] = rank;
  if (P.has_sinks) mark_sources(P.g, E.sinks, E.src_cell_start, E.src_sorted, E.src_occupied, ncell, nox, noy);
}

// ---------------------------------------------------------------------------
// K4: the neighbour pass.  Phase B of Simulation::step for one agent
// (lib.rs:259-347): radius query (location_hash_2d.rs:240-258), Zanlungo
// (zanlungo.rs:201-217).  Two forms with the same arithmetic and the same visiting order (cells
// x-major / y-minor, members of a cell in ascending id: the canonical order of SURVEY.md §8a'),
// so their results are bitwise equal:
//   k_step_tiled   cell lists staged in LDS by the workgroup, per-lane neighbour lists (fast path)
//   GatherSrc      cell lists read from global memory (exact for any grid and any agent state:
//                  clamped, aliased, overfull tiles); k_step_gather and the fallback in k_step_tiled
// ---------------------------------------------------------------------------
__device__ __forceinline__ float2 hlp_velocity(const GroupDev& grp, uint32_t id, const float2* pref,
                                               uint32_t i) {
  switch (grp.hlp_kind) {
    case CS_HLP_CONSTANT:
      return make_float2(grp.hvx, grp.hvy);
    case CS_HLP_ID_PARITY:  // rmf_crowdsim_viz/src/main.rs:26-29: even ids get -v
      return (id & 1u) ? make_float2(grp.hvx, grp.hvy) : make_float2(-grp.hvx, -grp.hvy);
    case CS_HLP_CALLBACK:
      return pref[i];
    default:
      return make_float2(0.0f, 0.0f);  // None: vel stays (0,0), lib.rs:263
  }
}

struct Own {
  float2 off, v, u;
  uint32_t id, gx, gy, slot;
};

// Members of one cell straight from the sorted global arrays.
struct GatherSrc {
  const AgentArrays& in;
  const uint32_t* __restrict__ cell_start;
  const GridDev& g;
  uint32_t self;
  __device__ __forceinline__ bool cell(long long x, long long y, uint32_t& b, uint32_t& e) const {
    if (x < 0 || y < 0) return false;  // signed_idx_to_data_idx, location_hash_2d.rs:74-85
    unsigned long long flat = (unsigned long long)x * g.nx + (unsigned long long)y;
    if (flat >= g.ncells) return false;
    b = cell_start[flat];
    e = cell_start[flat + 1];
    return true;
  }
  __device__ __forceinline__ bool is_self(uint32_t j) const { return j == self; }
  // offset of member j relative to the geometric cell (x, y) it was reached through;
  // a member stored through the aliasing of location_to_index sits in another cell
  __device__ __forceinline__ float2 off(uint32_t j, long long x, long long y) const {
    float2 o = in.off[j];
    if (y >= (long long)g.nx) {  // reached through an aliased flat index
      uint32_t c = in.cell[j];
      uint32_t ax = c / g.nx, ay = c - ax * g.nx;
      o.x += (float)((long long)ax - x) * g.cs;
      o.y += (float)((long long)ay - y) * g.cs;
    }
    return o;
  }
  __device__ __forceinline__ float2 vel(uint32_t j) const { return in.vel[j]; }
  __device__ __forceinline__ uint32_t id(uint32_t j) const { return in.id[j]; }
  // k-th member of [b, e) in ascending id: storage order inside a cell is arrival order
  __device__ __forceinline__ uint32_t ordered(uint32_t b, uint32_t e, uint32_t k, uint32_t& last,
                                              bool& have_last) const {
    uint32_t best = 0xFFFFFFFFu, bj = b;
    bool found = false;
    for (uint32_t j = b; j < e; ++j) {
      uint32_t idj = in.id[j];
      if ((!have_last || idj > last) && (!found || idj < best)) {
        best = idj;
        bj = j;
        found = true;
      }
    }
    last = best;
    have_last = true;
    return bj;
  }
};

// Returns the new velocity w = u + F/m of agent `o` (zanlungo.rs:201-217).  Generic form:
// every candidate is processed where it is found.  Used by the gather kernel.
template <class Src>
__device__ __forceinline__ void zanlungo_velocity(const Own& o, const GroupDev& grp, const GridDev& g,
                                                  long long lx, long long hx, long long ly,
                                                  long long hy, const Src& src, float& wx, float& wy,
                                                  bool& tti_zero) {
  const float r2 = grp.eyesight * grp.eyesight;
  const float R2 = grp.R * grp.R;
  // ---- compute_tti (zanlungo.rs:76-91): min over every neighbour in sight ----
  float T = f_inf();
  uint32_t n_back = 0;
  for (long long x = lx; x <= hx; ++x) {
    // own offset as seen from cell (x, y): p_j - p_i = off_j - (off_i - shift)
    const float oix = __builtin_fmaf(-(float)(x - (long long)o.gx), g.cs, o.off.x);
    for (long long y = ly; y <= hy; ++y) {
      uint32_t b, e;
      if (!src.cell(x, y, b, e)) continue;
      const float oiy = __builtin_fmaf(-(float)(y - (long long)o.gy), g.cs, o.off.y);
      for (uint32_t j = b; j < e; ++j) {
        if (src.is_self(j)) continue;  // lib.rs:284
        const float2 oj = src.off(j, x, y);
        const float rpx = oj.x - oix, rpy = oj.y - oiy;  // p_j - p_i
        const float d2 = __builtin_fmaf(rpx, rpx, rpy * rpy);
        if (!(d2 < r2)) continue;  // strict `<`, location_hash_2d.rs:251
        const float2 vj = src.vel(j);
        const float t = ttc_f32(vj.x - o.v.x, vj.y - o.v.y, rpx, rpy, d2, R2);
        T = (t < T) ? t : T;
        n_back += (src.id(j) < o.id) ? 1u : 0u;
      }
    }
  }
  tti_zero = (T == 0.0f);
  float fx = 0.0f, fy = 0.0f;
  if (T != f_inf()) {  // zanlungo.rs:211
    const ForceCtx fc = make_force_ctx(o.v.x, o.v.y, T, grp);
    for (long long x = lx; x <= hx; ++x) {
      const float oix = __builtin_fmaf(-(float)(x - (long long)o.gx), g.cs, o.off.x);
      for (long long y = ly; y <= hy; ++y) {
        uint32_t b, e;
        if (!src.cell(x, y, b, e)) continue;
        const float oiy = __builtin_fmaf(-(float)(y - (long long)o.gy), g.cs, o.off.y);
        uint32_t last = 0;
        bool have_last = false;
        for (uint32_t k = 0; k < e - b; ++k) {
          const uint32_t j = src.ordered(b, e, k, last, have_last);
          if (!(src.id(j) > o.id)) continue;  // self and smaller ids: weight 0
          const float2 oj = src.off(j, x, y);
          const float rpx = oj.x - oix, rpy = oj.y - oiy;
          const float d2 = __builtin_fmaf(rpx, rpx, rpy * rpy);
          if (!(d2 < r2)) continue;
          zanlungo_forward_force(rpx, rpy, d2, fc, fx, fy);
        }
      }
    }
    // Neighbours with a smaller id have weight 0: their term is (d/|d|) * 0 = 0, except
    // with t_i == 0, where 0 * A * |dv| / 0 = NaN (zanlungo.rs:163; SURVEY.md KAT-Z3).
    if (T == 0.0f && n_back > 0) {
      fx = f_nan();
      fy = f_nan();
    }
  }
  wx = __builtin_fmaf(fx, grp.inv_mass, o.u.x);  // recommended + force * (1/m), zanlungo.rs:216
  wy = __builtin_fmaf(fy, grp.inv_mass, o.u.y);
}

// get_bounds (location_hash_2d.rs:103-122) from a cell-relative position: cell range
// [gx + lo, gx + hi] for radius r.  Saturating like Rust's `as i64`.
__device__ __forceinline__ void cell_bounds(float o, float r, float inv_cs, long long g, long long& lo,
                                            long long& hi) {
  float fl = floorf((o - r) * inv_cs), fh = floorf((o + r) * inv_cs);
  lo = g + (long long)fminf(fmaxf(fl, -4e9f), 4e9f);
  hi = g + (long long)fminf(fmaxf(fh, -4e9f), 4e9f);
}

__device__ __forceinline__ void gather_agent(const StepParams& P, const AgentArrays& in,
                                             const uint32_t* __restrict__ cell_start, const Own& o,
                                             const GroupDev& grp, float& wx, float& wy, bool& tz) {
  long long lx, hx, ly, hy;
  cell_bounds(o.off.x, grp.eyesight, P.g.inv_cs, o.gx, lx, hx);
  cell_bounds(o.off.y, grp.eyesight, P.g.inv_cs, o.gy, ly, hy);
  // rows and cells that cannot exist: negative ones are rejected by the reference,
  // flat indices beyond the grid too; y may run past the stride (aliasing) up to the
  // last flat index
  lx = max(lx, 0ll);
  ly = max(ly, 0ll);
  hx = min(hx, (long long)(P.g.ncells / P.g.nx));
  hy = min(hy, (long long)P.g.ncells);
  GatherSrc src{in, cell_start, P.g, o.slot};
  zanlungo_velocity(o, grp, P.g, lx, hx, ly, hy, src, wx, wy, tz);
}

__global__ void __launch_bounds__(256) k_step_gather(StepParams P, AgentArrays in, EpilogueCtx E,
                                                     const uint32_t* __restrict__ cell_start,
                                                     const float2* __restrict__ pref) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (i >= E.ctr->n_alive) {
    E.out.cell[i] = CS_INVALID_CELL;  // slot beyond the live population
    return;
  }
  const uint32_t cell = in.cell[i];
  Own o;
  o.gx = cell / P.g.nx;
  o.gy = cell - o.gx * P.g.nx;
  if (P.g.tile && (o.gx < P.g.own_x0 || o.gx >= P.g.own_x1 || o.gy < P.g.own_y0 || o.gy >= P.g.own_y1)) {
    E.out.cell[i] = CS_INVALID_CELL;  // a ghost: its owner steps it
    return;
  }
  o.off = in.off[i];
  o.v = in.vel[i];
  o.id = in.id[i];
  o.slot = i;
  const uint32_t meta = in.meta[i];
  const GroupDev grp = E.groups[meta & 0xFFFFu];
  o.u = hlp_velocity(grp, o.id, pref, i);
  float wx = o.u.x, wy = o.u.y;  // NoLocalPlan: identity, no_local_plan.rs:9-17
  if (grp.lp_kind == 1u && o.off.x == o.off.x && o.off.y == o.off.y) {
    bool tz;
    gather_agent(P, in, cell_start, o, grp, wx, wy, tz);
    if (tz) atomicAdd(&E.ctr->n_tti_zero, 1u);
  }
  step_epilogue(P, E, i, o.gx, o.gy, o.off, o.id, meta, grp, wx, wy);
}

// ---------------------------------------------------------------------------
// Work decomposition for the tiled kernel: every workgroup gets up to 256
// consecutive agents of ONE grid row, so the cells it must see are (2h+1)
// contiguous ranges of the sorted arrays.  Built on the device after the scan.
// ---------------------------------------------------------------------------
#ifndef TILE_THREADS
#define TILE_THREADS 256  // agents (= threads) per workgroup of the tiled neighbour kernel
#endif

// A workgroup of the tiled kernel owns `nrows` consecutive grid rows x the cells [y0, y1].
// count != 0: a strip of one row cut at AGENT granularity (first, count <= TILE_THREADS).
// count == 0: a band window of nrows rows cut at CELL granularity (agents = the cells' members).
struct BlockDesc {
  uint32_t row0, nrows, y0, y1, first, count;
};

__global__ void __launch_bounds__(1024) k_build_blocks(GridDev g, const uint32_t* __restrict__ cell_start,
                                                       BlockDesc* __restrict__ desc, uint32_t desc_cap,
                                                       uint32_t* __restrict__ n_blocks,
                                                       Counters* __restrict__ ctr) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t s_carry, s_owned;
  if (threadIdx.x == 0) {
    s_carry = 0;
    s_owned = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // only the owned rows / columns get workgroups; ghosts are read, never stepped
  const uint32_t n_rows = g.own_x1 - g.own_x0;
  for (uint32_t base = 0; base < n_rows; base += blockDim.x) {
    uint32_t R = g.own_x0 + base + threadIdx.x;
    uint32_t first = 0, cnt = 0;
    if (R < g.own_x1) {
      first = cell_start[(unsigned long long)R * g.nx + g.own_y0];
      cnt = cell_start[(unsigned long long)R * g.nx + g.own_y1] - first;
    }
    uint32_t nb = (cnt + TILE_THREADS - 1u) / TILE_THREADS;
    uint32_t incl = wave_incl_scan(nb, lane);
    if (lane == 63) wsum[wave] = incl;
    if (cnt) atomicAdd(&s_owned, cnt);
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint32_t excl = s_carry + woff + incl - nb;
    for (uint32_t k = 0; k < nb; ++k) {
      if (excl + k < desc_cap) {
        BlockDesc d;
        d.row0 = R;
        d.nrows = 1;
        d.y0 = d.y1 = 0;
        d.first = first + k * TILE_THREADS;
        d.count = min((uint32_t)TILE_THREADS, cnt - k * TILE_THREADS);
        desc[excl + k] = d;
      }
    }
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_carry = excl + nb;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *n_blocks = min(s_carry, desc_cap);
    ctr->n_owned = s_owned;
    // runs after the scatter, before the step kernel, which writes one slot per sorted agent
    ctr->n_pending = cell_start[g.ncells];
  }
}

// Band windows: one workgroup of this builder per band of `rb` owned rows.  The inclusive
// prefix of the per-column agent counts goes to `prefix`; window w takes the columns whose
// first agent has band index in [w*target, (w+1)*target).  *n_blocks must be 0 on entry.
#define BAND_LDS_COLS 4096
__global__ void __launch_bounds__(256) k_build_bands(GridDev g, const uint32_t* __restrict__ cell_start,
                                                     uint32_t rb, uint32_t target,
                                                     uint32_t* __restrict__ prefix,
                                                     BlockDesc* __restrict__ desc, uint32_t desc_cap,
                                                     uint32_t* __restrict__ n_blocks,
                                                     Counters* __restrict__ ctr) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t s_carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t row0 = g.own_x0 + blockIdx.x * rb;
  const uint32_t nown = min(rb, g.own_x1 - row0);
  const uint32_t ncols = g.own_y1 - g.own_y0;
  // the prefix lives in LDS when the band is narrow enough (binary searches below), else in HBM
  __shared__ uint32_t s_incl[BAND_LDS_COLS];
  uint32_t* __restrict__ incl =
      ncols <= BAND_LDS_COLS ? s_incl : prefix + (unsigned long long)blockIdx.x * ncols;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < ncols; base += blockDim.x) {
    const uint32_t y = base + threadIdx.x;
    uint32_t c = 0;
    if (y < ncols)
      for (uint32_t r = 0; r < nown; ++r) {
        const unsigned long long q = (unsigned long long)(row0 + r) * g.nx + g.own_y0 + y;
        c += cell_start[q + 1] - cell_start[q];
      }
    const uint32_t w = wave_incl_scan(c, lane);
    if (lane == 63) wsum[wave] = w;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wave; ++k) woff += wsum[k];
    if (y < ncols) incl[y] = s_carry + woff + w;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_carry += woff + w;
    __syncthreads();
  }
  const uint32_t total = s_carry;
  if (threadIdx.x == 0) {
    if (total) atomicAdd(&ctr->n_owned, total);
    if (blockIdx.x == 0) ctr->n_pending = cell_start[g.ncells];
  }
  // smallest y with incl[y] >= v (ncols if none); incl is non-decreasing
  auto lb = [&](uint32_t v) {
    uint32_t lo = 0, hi = ncols;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (incl[mid] >= v) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  const uint32_t nw = (total + target - 1u) / target;
  for (uint32_t w = threadIdx.x; w < nw; w += blockDim.x) {
    const uint32_t lo = w * target, hi = lo + target;
    // first column whose first agent has band index >= lo (excl(y) = incl[y-1])
    const uint32_t a = lo == 0 ? 0u : lb(lo) + 1u;
    if (a >= ncols) continue;
    const uint32_t ea = a == 0 ? 0u : incl[a - 1];
    const uint32_t y0 = lb(ea + 1u);  // skip empty columns
    if (y0 >= ncols) continue;
    const uint32_t e0 = y0 == 0 ? 0u : incl[y0 - 1];
    if (e0 >= hi) continue;  // a single column jumped over this window
    // last column whose first agent has band index < hi
    const uint32_t b = lb(hi);  // columns > b have excl >= hi
    const uint32_t ylast = min(b, ncols - 1u);
    const uint32_t y1 = lb(incl[ylast]);  // drop trailing empty columns
    const uint32_t k = atomicAdd(n_blocks, 1u);
    if (k < desc_cap) {
      BlockDesc d;
      d.row0 = row0;
      d.nrows = nown;
      d.y0 = g.own_y0 + y0;
      d.y1 = g.own_y0 + y1;
      d.first = 0;
      d.count = 0;
      desc[k] = d;
    }
  }
}

#define TILE_MAX_ROWS 24  // owned rows (<= 8) + 2 * 8 ghost rows: eyesight up to 8 cells
#define TILE_MAX_OWN_ROWS 8

struct TileCfg {
  int h;                // ceil(max eyesight / cell)
  uint32_t agents_cap;  // LDS slots for staged agents
  uint32_t table_cap;   // u16 entries of the cell table
  uint32_t list_cap;    // neighbour-list entries per thread
  uint32_t debug;       // ablation switches for profiling (bench.py --debug), 0 in production
};

// Per-lane neighbour list entry: LDS slot of the neighbour and the cell it was found in
// relative to the own cell.  16-bit form (eyesight <= 1 cell, <= 4096 staged agents):
// slot | (dy+1) << 12 | (dx+1) << 14.  32-bit form: slot | dx << 16 | dy << 24 (signed bytes).
template <bool E16>
struct ListEntry;
template <>
struct ListEntry<true> {
  typedef unsigned short T;
  static __device__ __forceinline__ T make(uint32_t j, int dx, int dy) {
    return (T)(j | ((uint32_t)(dy + 1) << 12) | ((uint32_t)(dx + 1) << 14));
  }
  static __device__ __forceinline__ void unpack(T e, uint32_t& j, float& fdx, float& fdy) {
    j = e & 0xFFFu;
    fdx = (float)(int)((e >> 14) & 3u) - 1.0f;
    fdy = (float)(int)((e >> 12) & 3u) - 1.0f;
  }
};
template <>
struct ListEntry<false> {
  typedef uint32_t T;
  static __device__ __forceinline__ T make(uint32_t j, int dx, int dy) {
    return j | ((uint32_t)(dx & 0xFF) << 16) | ((uint32_t)(dy & 0xFF) << 24);
  }
  static __device__ __forceinline__ void unpack(T e, uint32_t& j, float& fdx, float& fdy) {
    j = e & 0xFFFFu;
    fdx = (float)((int)(e << 8) >> 24);
    fdy = (float)((int)e >> 24);
  }
};

// K4 (tiled form): one workgroup = one BlockDesc = up to 256 consecutive agents of one
// grid row.  The (2h+1) cell-row segments around the strip are staged in LDS once (members
// of a cell in ascending id), then every thread runs the neighbour pass of its agent:
//   1. distance filter over the (2h+1)^2 cells around the agent -> compacted list in LDS
//   2. time-to-collision over the list (min -> t_i); the neighbours with right of way
//      (larger id) are remembered in a 64-bit lane mask
//   3. forces over the marked entries, in list (= canonical) order
// Lists are bounded (list_cap <= 64); when any lane of a wave fills up, the wave processes
// what it has and carries on (pass 3 then re-runs the filter, since the list was recycled).
template <bool E16>
__global__ void __launch_bounds__(TILE_THREADS) k_step_tiled(
    StepParams P, AgentArrays in, EpilogueCtx E, const uint32_t* __restrict__ cell_start,
    const float2* __restrict__ pref, const BlockDesc* __restrict__ desc,
    const uint32_t* __restrict__ n_blocks, TileCfg cfg) {
  typedef ListEntry<E16> LE;
  typedef typename LE::T entry_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_g0[TILE_MAX_ROWS], s_base[TILE_MAX_ROWS + 1];
  {  // slots beyond the live population must not look alive to the next scatter
    const uint32_t t = E.ctr->n_alive + blockIdx.x * TILE_THREADS + threadIdx.x;
    if (t < P.n) E.out.cell[t] = CS_INVALID_CELL;
  }
  if (blockIdx.x >= *n_blocks) return;
  const BlockDesc d = desc[blockIdx.x];
  const GridDev g = P.g;
  const int tid = threadIdx.x;
  float2* __restrict__ s_off = reinterpret_cast<float2*>(smem);
  float2* __restrict__ s_vel = s_off + cfg.agents_cap;
  uint32_t* __restrict__ s_id = reinterpret_cast<uint32_t*>(s_vel + cfg.agents_cap);
  entry_t* __restrict__ s_list = reinterpret_cast<entry_t*>(s_id + cfg.agents_cap);  // [cap][256]
  // ids in arrival order, needed only while staging: they borrow the (still unused) list area
  uint32_t* __restrict__ s_idtmp = reinterpret_cast<uint32_t*>(s_list);
  unsigned short* __restrict__ s_tab =
      reinterpret_cast<unsigned short*>(s_list + cfg.list_cap * TILE_THREADS);

  // ---- geometry of the owned rows x cells and their halo ----
  __shared__ uint32_t s_rfirst[TILE_MAX_OWN_ROWS], s_rpref[TILE_MAX_OWN_ROWS + 1];
  const int n_rows = (int)(g.ncells / g.nx);
  const int row0 = (int)d.row0, nown = (int)d.nrows;
  const bool strip = d.count != 0;  // cut at agent granularity (k_build_blocks), else a band window
  int ylo, yhi;
  if (strip) {
    const uint32_t c_lo = in.cell[d.first], c_hi = in.cell[d.first + d.count - 1];
    ylo = (int)(c_lo - (uint32_t)row0 * g.nx);
    yhi = (int)(c_hi - (uint32_t)row0 * g.nx);
  } else {
    ylo = (int)d.y0;
    yhi = (int)d.y1;
  }
  const int sy0 = max(ylo - cfg.h, 0), sy1 = min(yhi + cfg.h, (int)g.nx - 1);
  const int W1 = sy1 - sy0 + 2;
  const int r0 = max(row0 - cfg.h, 0), r1 = min(row0 + nown - 1 + cfg.h, n_rows - 1);
  const int nr = r1 - r0 + 1;
  if (tid < nr) {
    unsigned long long rowbase = (unsigned long long)(r0 + tid) * g.nx;
    uint32_t g0 = cell_start[rowbase + sy0], g1 = cell_start[rowbase + sy1 + 1];
    s_g0[tid] = g0;
    s_base[tid + 1] = g1 - g0;
  }
  if (tid >= 64 && tid < 64 + nown) {  // the owned agents of each owned row (another wave)
    const int r = tid - 64;
    if (strip) {
      s_rfirst[0] = d.first;
      s_rpref[1] = d.count;
    } else {
      const unsigned long long rowbase = (unsigned long long)(row0 + r) * g.nx;
      const uint32_t f = cell_start[rowbase + ylo];
      s_rfirst[r] = f;
      s_rpref[r + 1] = cell_start[rowbase + yhi + 1] - f;
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    s_base[0] = 0;
    for (int k = 0; k < nr; ++k) {
      uint32_t c = s_base[k + 1];
      s_base[k + 1] = acc + c;
      acc += c;
    }
    acc = 0;
    s_rpref[0] = 0;
    for (int k = 0; k < nown; ++k) {
      uint32_t c = s_rpref[k + 1];
      s_rpref[k + 1] = acc + c;
      acc += c;
    }
  }
  __syncthreads();
  const uint32_t S = s_base[nr];
  const uint32_t n_own = s_rpref[nown];
  const bool tiled_ok = S <= cfg.agents_cap && S < (E16 ? 4096u : 65535u) &&
                        (uint32_t)(nr * W1) <= cfg.table_cap;

  if (tiled_ok) {
    // cell table: first LDS slot of every staged cell (+ one end marker per row)
    for (int k = 0; k < nr; ++k) {
      const unsigned long long rowbase = (unsigned long long)(r0 + k) * g.nx + sy0;
      for (int y = tid; y < W1; y += TILE_THREADS)
        s_tab[k * W1 + y] = (unsigned short)(s_base[k] + (cell_start[rowbase + y] - s_g0[k]));
    }
    // ids first, in arrival (= global) order, so the rank of an agent inside its cell can be
    // counted from LDS
    for (uint32_t s = tid; s < S; s += TILE_THREADS) {
      int k = 0;
      while (s >= s_base[k + 1]) ++k;
      s_idtmp[s] = in.id[s_g0[k] + (s - s_base[k])];
    }
    __syncthreads();
    // agents, each placed at its cell's first slot + its rank by id inside the cell
    for (uint32_t s = tid; s < S; s += TILE_THREADS) {
      int k = 0;
      while (s >= s_base[k + 1]) ++k;
      const uint32_t j = s_g0[k] + (s - s_base[k]);
      const int cy = (int)(in.cell[j] - (uint32_t)(r0 + k) * g.nx) - sy0;
      const uint32_t idj = s_idtmp[s];
      const unsigned short* t = s_tab + k * W1 + cy;
      const uint32_t tb = t[0], te = t[1];
      uint32_t rank = 0;
      if (cfg.debug & 2u) rank = s - tb;
      else
        for (uint32_t q = tb; q < te; ++q) rank += (s_idtmp[q] < idj) ? 1u : 0u;
      const uint32_t slot = tb + rank;
      s_off[slot] = in.off[j];
      s_vel[slot] = in.vel[j];
      s_id[slot] = idj;
    }
  }
  __syncthreads();

  // ---- one thread = one agent; idle lanes keep the wave-uniform loops going.  A window that
  // holds more than TILE_THREADS agents (dense spots) is walked in chunks. ----
  for (uint32_t chunk = 0; chunk < n_own; chunk += TILE_THREADS) {
  const uint32_t a_idx = chunk + (uint32_t)tid;
  const bool active = a_idx < n_own;
  int own_r = 0;
  if (active)
    while (own_r + 1 < nown && a_idx >= s_rpref[own_r + 1]) ++own_r;
  const int R = row0 + own_r;
  const uint32_t i = active ? s_rfirst[own_r] + (a_idx - s_rpref[own_r]) : s_rfirst[0];
  const uint32_t cell = in.cell[i];
  Own o;
  o.gx = (uint32_t)R;
  o.gy = cell - (uint32_t)R * g.nx;
  o.off = in.off[i];
  o.v = in.vel[i];
  o.id = in.id[i];
  o.slot = i;
  const uint32_t meta = in.meta[i];
  const GroupDev grp = E.groups[meta & 0xFFFFu];
  o.u = hlp_velocity(grp, o.id, pref, i);
  float wx = o.u.x, wy = o.u.y;  // NoLocalPlan: identity, no_local_plan.rs:9-17
  const bool zan = active && grp.lp_kind == 1u && o.off.x == o.off.x && o.off.y == o.off.y;

  // an agent whose offset left its cell (clamped below the grid, aliased above it), or whose
  // reach crosses the row stride (where the reference aliases into the next row), needs
  // cells the strip did not stage: exact gather path
  int lx = 1, hx = 0, ly = 1, hy = 0;  // empty by default
  bool use_tile = false;
  if (zan) {
    const float slack = 0.01f * g.cs;
    const bool regular = o.off.x >= -slack && o.off.x <= g.cs + slack && o.off.y >= -slack &&
                         o.off.y <= g.cs + slack;
    if (tiled_ok && regular) {
      long long blx, bhx, bly, bhy;
      cell_bounds(o.off.x, grp.eyesight, g.inv_cs, 0, blx, bhx);
      cell_bounds(o.off.y, grp.eyesight, g.inv_cs, 0, bly, bhy);
      use_tile = (long long)o.gy + bhy < (long long)g.nx;
      // f32 rounding at a cell edge can ask for one cell beyond ceil(r / cell); that cell
      // lies entirely out of reach
      lx = (int)max(blx, (long long)-cfg.h);
      hx = (int)min(bhx, (long long)cfg.h);
      ly = (int)max(bly, (long long)-cfg.h);
      hy = (int)min(bhy, (long long)cfg.h);
    }
  }
  if (zan && !use_tile) {
    bool tz;
    gather_agent(P, in, cell_start, o, grp, wx, wy, tz);
    if (tz) atomicAdd(&E.ctr->n_tti_zero, 1u);
  }

  if (tiled_ok && !(cfg.debug & 1u)) {  // block-uniform; lanes without tile work idle through
    const bool mine = zan && use_tile;
    if (!mine) {
      lx = 1; hx = 0; ly = 1; hy = 0;
    }
    const float r2 = grp.eyesight * grp.eyesight;
    const float R2 = grp.R * grp.R;
    const int own_row = R - r0;
    // own LDS slot, to skip self without touching ids
    uint32_t self_slot = 0xFFFFFFFFu;
    if (mine) {
      const unsigned short* t = s_tab + own_row * W1 + ((int)o.gy - sy0);
      for (uint32_t j = t[0]; j < t[1]; ++j)
        if (s_id[j] == o.id) self_slot = j;
    }
    entry_t* __restrict__ my_list = s_list + tid;
    const uint32_t CAP = cfg.list_cap;
    uint32_t cnt = 0, n_back = 0;
    unsigned long long fwd = 0;  // bit k: list entry k has right of way over this agent
    float T = f_inf();
    bool flushed = false;  // wave-uniform

    auto rel = [&](entry_t ent, uint32_t& j, float& rpx, float& rpy) {
      float fdx, fdy;
      LE::unpack(ent, j, fdx, fdy);
      // own offset as seen from the neighbour's cell: p_j - p_i = off_j - (off_i - shift)
      const float oix = __builtin_fmaf(-fdx, g.cs, o.off.x);
      const float oiy = __builtin_fmaf(-fdy, g.cs, o.off.y);
      const float2 oj = s_off[j];
      rpx = oj.x - oix;
      rpy = oj.y - oiy;
    };
    // pass 2 over the first `n` entries of the list: t_i and the right-of-way mask.
    // Two entries per trip: their LDS reads are issued together, so one latency covers both.
    auto run_ttc = [&](uint32_t n) {
      unsigned long long m = 0;
      uint32_t k = 0;
      for (; k + 1 < n; k += 2) {
        const entry_t e0 = my_list[k * TILE_THREADS], e1 = my_list[(k + 1) * TILE_THREADS];
        uint32_t j0, j1;
        float fdx0, fdy0, fdx1, fdy1;
        LE::unpack(e0, j0, fdx0, fdy0);
        LE::unpack(e1, j1, fdx1, fdy1);
        const float2 oj0 = s_off[j0], oj1 = s_off[j1];
        const float2 vj0 = s_vel[j0], vj1 = s_vel[j1];
        const uint32_t id0 = s_id[j0], id1 = s_id[j1];
        const float rpx0 = oj0.x - __builtin_fmaf(-fdx0, g.cs, o.off.x);
        const float rpy0 = oj0.y - __builtin_fmaf(-fdy0, g.cs, o.off.y);
        const float rpx1 = oj1.x - __builtin_fmaf(-fdx1, g.cs, o.off.x);
        const float rpy1 = oj1.y - __builtin_fmaf(-fdy1, g.cs, o.off.y);
        const float t0 = ttc_f32(vj0.x - o.v.x, vj0.y - o.v.y, rpx0, rpy0,
                                 __builtin_fmaf(rpx0, rpx0, rpy0 * rpy0), R2);
        const float t1 = ttc_f32(vj1.x - o.v.x, vj1.y - o.v.y, rpx1, rpy1,
                                 __builtin_fmaf(rpx1, rpx1, rpy1 * rpy1), R2);
        T = (t0 < T) ? t0 : T;
        T = (t1 < T) ? t1 : T;
        const bool f0 = id0 > o.id, f1 = id1 > o.id;
        m |= (f0 ? (1ull << k) : 0ull) | (f1 ? (2ull << k) : 0ull);
        n_back += (f0 ? 0u : 1u) + (f1 ? 0u : 1u);
      }
      if (k < n) {
        uint32_t j;
        float rpx, rpy;
        rel(my_list[k * TILE_THREADS], j, rpx, rpy);
        const float2 vj = s_vel[j];
        const uint32_t idj = s_id[j];
        const float t = ttc_f32(vj.x - o.v.x, vj.y - o.v.y, rpx, rpy, __builtin_fmaf(rpx, rpx, rpy * rpy), R2);
        T = (t < T) ? t : T;
        const bool f = idj > o.id;
        m |= f ? (1ull << k) : 0ull;
        n_back += f ? 0u : 1u;
      }
      fwd = m;
    };
    // pass 3: forces of the listed neighbours selected by `mask`, in list order
    float fx = 0.0f, fy = 0.0f;
    ForceCtx fc;
    auto run_force = [&](unsigned long long mask) {
      while (mask) {
        const int k = __ffsll((long long)mask) - 1;
        mask &= mask - 1ull;
        uint32_t j;
        float rpx, rpy;
        rel(my_list[k * TILE_THREADS], j, rpx, rpy);
        const float d2 = __builtin_fmaf(rpx, rpx, rpy * rpy);
        zanlungo_forward_force(rpx, rpy, d2, fc, fx, fy);
      }
    };
    // pass 1: the filter.  FORCE = false collects every neighbour in sight, FORCE = true only
    // those with a larger id (used when the list had to be recycled).
    auto sweep = [&](bool FORCE) {
      for (int dx = -cfg.h; dx <= cfg.h; ++dx) {
        const int rr = own_row + dx;  // staged row index (per lane: lanes own different rows)
        const bool x_in = dx >= lx && dx <= hx && rr >= 0 && rr < nr;
        const float oix = __builtin_fmaf(-(float)dx, g.cs, o.off.x);
        for (int dy = -cfg.h; dy <= cfg.h; ++dy) {
          const int cy = (int)o.gy + dy - sy0;
          uint32_t j = 0, e = 0;
          if (x_in && dy >= ly && dy <= hy && cy >= 0 && cy < W1 - 1) {
            const unsigned short* t = s_tab + rr * W1 + cy;
            j = t[0];
            e = t[1];
          }
          const float oiy = __builtin_fmaf(-(float)dy, g.cs, o.off.y);
          if (cfg.debug & 16u) j = e;
          while (__any(j < e)) {
            // two candidates per trip while there is room for both: one LDS latency for two tests
            while (j + 1 < e && cnt + 1 < CAP) {
              const float2 oa = s_off[j], ob = s_off[j + 1];
              const float ax = oa.x - oix, ay = oa.y - oiy, bx = ob.x - oix, by = ob.y - oiy;
              bool ta = __builtin_fmaf(ax, ax, ay * ay) < r2 && j != self_slot;
              bool tb = __builtin_fmaf(bx, bx, by * by) < r2 && j + 1 != self_slot;
              if (FORCE) {
                if (ta) ta = s_id[j] > o.id;
                if (tb) tb = s_id[j + 1] > o.id;
              }
              // branch-free append: a rejected entry is overwritten by the next store
              my_list[cnt * TILE_THREADS] = LE::make(j, dx, dy);
              cnt += ta ? 1u : 0u;
              my_list[cnt * TILE_THREADS] = LE::make(j + 1, dx, dy);
              cnt += tb ? 1u : 0u;
              j += 2;
            }
            while (j < e && cnt < CAP) {
              const float2 oj = s_off[j];
              const float rpx = oj.x - oix, rpy = oj.y - oiy;
              const float d2 = __builtin_fmaf(rpx, rpx, rpy * rpy);
              bool take = d2 < r2 && j != self_slot;  // strict `<`, location_hash_2d.rs:251
              if (FORCE && take) take = s_id[j] > o.id;
              if (take) {
                my_list[cnt * TILE_THREADS] = LE::make(j, dx, dy);
                ++cnt;
              }
              ++j;
            }
            if (__any(cnt >= CAP)) {  // some lane is full: everyone drains
              if (FORCE) {
                run_force(cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull));
              } else {
                run_ttc(cnt);
                flushed = true;
              }
              cnt = 0;
            }
          }
        }
      }
    };

    sweep(false);
    if (cfg.debug & 8u) cnt = 0;
    run_ttc(cnt);
    const bool tz = (T == 0.0f);
    if (mine && T != f_inf()) fc = make_force_ctx(o.v.x, o.v.y, T, grp);
    else fc = ForceCtx{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // lanes with t_i = inf take no force (zanlungo.rs:211)
    if (!__any(flushed)) {
      run_force((T != f_inf()) ? fwd : 0ull);
    } else {
      cnt = 0;
      if (T == f_inf()) {
        lx = 1; hx = 0;  // nothing to collect
      }
      sweep(true);
      run_force(cnt >= 64u ? ~0ull : ((1ull << cnt) - 1ull));
    }
    if (mine) {
      if (T != f_inf()) {
        // Neighbours with a smaller id have weight 0: their term is (d/|d|) * 0 = 0, except
        // with t_i == 0, where 0 * A * |dv| / 0 = NaN (zanlungo.rs:163; SURVEY.md KAT-Z3).
        if (tz && n_back > 0) {
          fx = f_nan();
          fy = f_nan();
        }
      } else {
        fx = 0.0f;
        fy = 0.0f;
      }
      wx = __builtin_fmaf(fx, grp.inv_mass, o.u.x);  // recommended + force * (1/m), zanlungo.rs:216
      wy = __builtin_fmaf(fy, grp.inv_mass, o.u.y);
      if (tz) atomicAdd(&E.ctr->n_tti_zero, 1u);
    }
  }
  if (active && !(cfg.debug & 4u)) step_epilogue(P, E, i, o.gx, o.gy, o.off, o.id, meta, grp, wx, wy);
  }  // chunk
}

// ---------------------------------------------------------------------------
// K7: halo exchange of a tile (DESIGN.md "Tiles").  A record is what a neighbour tile needs
// to see an agent: cell-relative state + its GLOBAL cell.  Ownership is decided by cell
// alone, so migrants and ghosts travel the same way: everything within 2*halo cells of a
// shared edge (the ghost ring itself holds the agents that just walked out) is sent.
// Buffer layout: record 0 is the header (word 0 = record count), records follow.
// ---------------------------------------------------------------------------
struct HaloRecord {
  float ox, oy, vx, vy;
  uint32_t id, meta, gcx, gcy;
};
static_assert(sizeof(HaloRecord) == CS_HALO_RECORD_BYTES, "halo record layout");

__device__ __forceinline__ void halo_append(HaloRecord* __restrict__ buf, uint32_t cap, const HaloRecord& r,
                                            Counters* ctr) {
  uint32_t k = atomicAdd(reinterpret_cast<uint32_t*>(buf), 1u);
  if (k < cap) buf[k + 1] = r;
  else atomicAdd(&ctr->n_halo_overflow, 1u);
}

__global__ void k_halo_pack(GridDev g, AgentArrays a, uint32_t n_ub, uint32_t axis, uint32_t band,
                            HaloRecord* __restrict__ send_lo, HaloRecord* __restrict__ send_hi,
                            uint32_t cap, Counters* __restrict__ ctr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_ub || i >= ctr->n_pending) return;
  const uint32_t c = a.cell[i];
  if (c == CS_INVALID_CELL) return;
  const uint32_t cx = c / g.nx, cy = c - cx * g.nx;
  const uint32_t q = axis == 0 ? cx : cy;
  const uint32_t extent = axis == 0 ? g.ncells / g.nx : g.nx;
  const bool lo = send_lo != nullptr && q < band;
  const bool hi = send_hi != nullptr && q + band >= extent;
  if (!lo && !hi) return;
  HaloRecord r;
  const float2 o = a.off[i], v = a.vel[i];
  r.ox = o.x; r.oy = o.y; r.vx = v.x; r.vy = v.y;
  r.id = a.id[i];
  r.meta = a.meta[i];
  r.gcx = g.org_x + cx;
  r.gcy = g.org_y + cy;
  if (lo) halo_append(send_lo, cap, r, ctr);
  if (hi) halo_append(send_hi, cap, r, ctr);
}

__global__ void k_halo_unpack(GridDev g, AgentArrays a, uint32_t slot_cap,
                              const HaloRecord* __restrict__ recv, uint32_t cap,
                              uint32_t* __restrict__ cell_count, Counters* __restrict__ ctr) {
  const uint32_t n = min(reinterpret_cast<const uint32_t*>(recv)[0], cap);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const HaloRecord r = recv[i + 1];
  const long long lx = (long long)r.gcx - (long long)g.org_x, ly = (long long)r.gcy - (long long)g.org_y;
  const long long n_rows = g.ncells / g.nx;
  if (lx < 0 || ly < 0 || lx >= n_rows || ly >= (long long)g.nx) return;  // not in my ring
  const uint32_t slot = atomicAdd(&ctr->n_pending, 1u);
  if (slot >= slot_cap) {
    atomicAdd(&ctr->n_halo_overflow, 1u);
    return;
  }
  const uint32_t cell = (uint32_t)lx * g.nx + (uint32_t)ly;
  a.off[slot] = make_float2(r.ox, r.oy);
  a.vel[slot] = make_float2(r.vx, r.vy);
  a.id[slot] = r.id;
  a.meta[slot] = r.meta;
  a.cell[slot] = cell;
  a.rank[slot] = atomicAdd(&cell_count[cell], 1u);
}

// ---------------------------------------------------------------------------
// K6: spawn.  One block; sinks in ascending handle order.  A sink spawns ONE
// agent at its source iff its generator asked for > 0 and nobody stood within
// 0.4 of the source at the end of the previous step (lib.rs:199-254).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_spawn(AgentArrays a, uint32_t n_slots, uint32_t cap,
                                                const SinkDev* __restrict__ sinks,
                                                const uint32_t* __restrict__ want,  // per sink slot
                                                uint32_t n_sinks, uint32_t n_want,
                                                const uint32_t* __restrict__ src_occupied,
                                                uint32_t* __restrict__ cell_count, uint32_t first_id,
                                                uint32_t* __restrict__ spawned_slots,
                                                Counters* __restrict__ ctr) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t base = 0; base < n_sinks; base += blockDim.x) {
    uint32_t s = base + threadIdx.x;
    uint32_t flag = 0;
    if (s < n_sinks)
      flag = (want[s] > 0 && src_occupied[s] == 0 && sinks[s].src_cell != CS_INVALID_CELL) ? 1u : 0u;
    uint32_t incl = wave_incl_scan(flag, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint32_t excl = s_carry + woff + incl - flag;
    if (flag) {
      uint32_t slot = n_slots + excl;
      if (slot < cap) {
        const SinkDev& sk = sinks[s];
        a.off[slot] = make_float2(sk.src_ox, sk.src_oy);
        a.vel[slot] = make_float2(0.0f, 0.0f);
        a.id[slot] = first_id + excl;
        a.cell[slot] = sk.src_cell;
        a.meta[slot] = sk.group;  // next_waypoint = 0
        a.rank[slot] = atomicAdd(&cell_count[sk.src_cell], 1u);
        spawned_slots[excl] = s;
      }
    }
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) s_carry = excl + flag;
    __syncthreads();
  }
  // the host reserved n_want slots; the ones no sink used must not look alive
  for (uint32_t k = s_carry + threadIdx.x; k < n_want; k += blockDim.x)
    if (n_slots + k < cap) a.cell[n_slots + k] = CS_INVALID_CELL;
  if (threadIdx.x == 0) ctr->n_spawned = s_carry;
}

// ---------------------------------------------------------------------------
// radius query against the sorted state (SpatialIndex::get_neighbours_in_radius,
// location_hash_2d.rs:240-258).  One block, cells in x-major / y-minor order;
// used by cs_query_radius (host API), not by the step.
// ---------------------------------------------------------------------------
__global__ void k_query_radius(GridDev g, AgentArrays a, const uint32_t* __restrict__ cell_start,
                               long long lx, long long hx, long long ly, long long hy,
                               uint32_t qcx, uint32_t qcy, float qox, float qoy, float r,
                               uint32_t* __restrict__ out_ids, uint32_t out_cap,
                               uint32_t* __restrict__ out_count) {
  // serial over cells (order matters), parallel inside a cell is not needed: tiny
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t n = 0;
  for (long long x = lx; x <= hx; ++x) {
    for (long long y = ly; y <= hy; ++y) {
      if (x < 0 || y < 0) continue;
      unsigned long long flat = (unsigned long long)x * g.nx + (unsigned long long)y;
      if (flat >= g.ncells) continue;
      uint32_t b = cell_start[flat], e = cell_start[flat + 1];
      // ascending id inside the cell: selection order over the (small) cell
      uint32_t last = 0;
      bool first = true;
      for (uint32_t k = b; k < e; ++k) {
        uint32_t best = 0xFFFFFFFFu, bj = 0;
        bool found = false;
        for (uint32_t j = b; j < e; ++j) {
          uint32_t idj = a.id[j];
          if ((first || idj > last) && (!found || idj < best)) {
            best = idj;
            bj = j;
            found = true;
          }
        }
        if (!found) break;
        last = best;
        first = false;
        uint32_t cj = a.cell[bj];
        uint32_t ax = cj / g.nx, ay = cj - ax * g.nx;
        float2 oj = a.off[bj];
        float rx = (float)((long long)ax - (long long)qcx) * g.cs + (oj.x - qox);
        float ry = (float)((long long)ay - (long long)qcy) * g.cs + (oj.y - qoy);
        if (sqrtf(rx * rx + ry * ry) < r) {
          if (n < out_cap) out_ids[n] = best;
          ++n;
        }
      }
    }
  }
  *out_count = n;
}

// ===========================================================================
// host side
// ===========================================================================
namespace {

inline uint64_t sat_usize(double v) {
  if (!(v > 0.0)) return 0;
  if (v >= 18446744073709551615.0) return UINT64_MAX;
  return (uint64_t)v;
}

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// Seeded stand-in for PoissonCrowd (source_sink.rs:75-82, whose thread_rng is
// unseedable): Knuth's product method on uniforms from splitmix64(seed, step,
// draw).  Integer mixing + IEEE f64 products: the same sequence on any host.
inline uint64_t poisson_seeded(uint64_t seed, uint64_t step, double mean) {
  if (!(mean > 0.0)) return 0;
  double limit = std::exp(-mean), prod = 1.0;
  uint64_t k = 0;
  for (uint64_t draw = 0; draw < 1000000; ++draw) {
    uint64_t r = splitmix64(seed ^ splitmix64(step * 0x100000001B3ull + draw));
    double u = (double)((r >> 11) + 1) * (1.0 / 9007199254740993.0);
    prod *= u;
    if (prod <= limit) break;
    ++k;
  }
  return k;
}

struct HostSink {
  cs_source_sink_desc d;
  std::vector<double> waypoints;
  uint64_t calls = 0;
  bool alive = true;
  uint32_t group = 0;
};

struct HostGroup {
  uint32_t hlp, lp;
  double eyesight;
  int32_t sink;
};

}  // namespace

#define HIP_OK(call)                                                                          \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #call;              \
      return 90;                                                                              \
    }                                                                                         \
  } while (0)

struct cs_engine {
  // grid (LocationHash2D::new, location_hash_2d.rs:33-51)
  cs_grid_desc grid;
  uint64_t nx = 0, ny = 0, ncells = 0;  // LOCAL grid: nx = row stride (columns), ny = rows
  uint64_t gnx = 0, gny = 0;            // global grid (== local unless this is a tile)
  GridDev gdev;
  bool tile = false;
  uint32_t halo_cells = 0;
  struct HaloDir {
    HaloRecord* send = nullptr;
    HaloRecord* recv = nullptr;
    uint32_t cap = 0;
  } halo[4];
  int device = 0;
  uint32_t flags = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string error;
  std::string backend;
  bool poisoned = false;

  // agent buffers: buf[cur] holds the state, buf[cur^1] is the scatter / step target
  AgentArrays buf[2] = {};
  int cur = 0;
  uint64_t cap = 0;
  uint32_t n_slots = 0;     // slots in use in buf[cur]
  bool sorted = true;       // buf[cur] is in cell order, cell_start/n_alive match it
  bool hist_valid = false;  // cell_count + rank describe buf[cur]
  bool occ_valid = false;   // src_occupied describes the current positions
  float2* pref = nullptr;   // callback-HLP velocities in sorted order
  uint32_t* cell_count = nullptr;
  uint32_t* cell_start = nullptr;
  uint32_t* block_totals = nullptr;
  uint32_t n_scan_blocks = 0;
  Counters* ctr = nullptr;       // device
  Counters* ctr_host = nullptr;  // pinned
  uint2* destroyed = nullptr;
  uint32_t destroyed_cap = 0;
  WpEvent* wp_events = nullptr;
  uint32_t wp_events_cap = 0;
  BlockDesc* blk_desc = nullptr;  // work decomposition of the tiled neighbour kernel
  uint32_t blk_desc_cap = 0;
  uint32_t* n_blocks_dev = nullptr;
  double max_eyesight = 0.0;
  uint32_t tile_blocks_per_cu = 4;  // LDS budget target of the tiled kernel (tuning knobs)
  uint32_t tile_list_cap = 0;       // 0 = derive from the budget
  uint32_t tile_agents_slack = 0;
  uint32_t tile_rows = 2;      // owned rows per workgroup (1 = strips cut at agent granularity)
  uint32_t tile_target = 224;  // agents per band window
  uint32_t* band_prefix = nullptr;

  // planners as data
  std::vector<cs_zanlungo_params> lp_params;
  std::vector<uint32_t> lp_kinds;
  std::vector<cs_hlp_desc> hlps;
  std::vector<HostGroup> groups;
  GroupDev* groups_dev = nullptr;
  bool groups_dirty = true;
  bool any_callback_hlp = false;

  // source-sinks: slot == handle (registry.rs:16-21 hands out ascending integers)
  std::vector<HostSink> sinks;
  SinkDev* sinks_dev = nullptr;
  double* waypoints_dev = nullptr;
  uint32_t* src_cell_start = nullptr;
  uint32_t* src_sorted = nullptr;
  uint32_t* src_occupied = nullptr;
  uint32_t* want_dev = nullptr;
  uint32_t* want_host = nullptr;  // pinned: the H2D copy of a step may still be in flight at return
  uint32_t* spawned_slots_dev = nullptr;
  bool sinks_dirty = true;
  uint32_t n_live_sinks = 0;
  bool spawn_committed = false;          // tile engines: Phase A ran through cs_spawn_probe/commit
  uint32_t committed_spawns = 0;
  SpawnRecord* spawn_rec_dev = nullptr;
  uint32_t spawn_rec_cap = 0;

  uint64_t next_id = 0;  // last_alloc_agent_id, lib.rs:83
  uint64_t n_alive_host = 0;
  std::vector<cs_event> events;
  bool record_events = true;

  // per-kernel hipEvent timing
  uint32_t profiling = 0;  // bitmask of CS_K_* kernels to time
  bool prof_open = false;
  struct Timed {
    hipEvent_t a, b;
    uint32_t k;
  };
  std::vector<Timed> timed;
  std::vector<hipEvent_t> event_pool;
  double prof_ms[CS_K_COUNT] = {};
  uint64_t prof_n[CS_K_COUNT] = {};

  // ---- memory ----
  int alloc_arrays(AgentArrays& a, uint64_t n) {
    HIP_OK(hipMalloc(&a.off, n * sizeof(float2)));
    HIP_OK(hipMalloc(&a.vel, n * sizeof(float2)));
    HIP_OK(hipMalloc(&a.id, n * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&a.cell, n * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&a.meta, n * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&a.rank, n * sizeof(uint32_t)));
    return 0;
  }
  void free_arrays(AgentArrays& a) {
    hipFree(a.off); hipFree(a.vel); hipFree(a.id); hipFree(a.cell); hipFree(a.meta); hipFree(a.rank);
    a = AgentArrays{};
  }
  int reserve(uint64_t need) {
    for (int d = 0; d < 4; ++d) need += 2ull * halo[d].cap;  // last step's ghosts + this step's arrivals
    if (need <= cap) return 0;
    uint64_t ncap = std::max<uint64_t>(need, std::max<uint64_t>(1024, cap * 2));
    if (ncap >= 0xFFFFFFF0ull) {
      error = "agent capacity exceeds 32-bit slots";
      return 4;
    }
    HIP_OK(hipStreamSynchronize(stream));
    for (int b = 0; b < 2; ++b) {
      AgentArrays na{};
      if (int rc = alloc_arrays(na, ncap)) return rc;
      if (cap && b == cur && n_slots) {
        HIP_OK(hipMemcpy(na.off, buf[b].off, n_slots * sizeof(float2), hipMemcpyDeviceToDevice));
        HIP_OK(hipMemcpy(na.vel, buf[b].vel, n_slots * sizeof(float2), hipMemcpyDeviceToDevice));
        HIP_OK(hipMemcpy(na.id, buf[b].id, n_slots * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        HIP_OK(hipMemcpy(na.cell, buf[b].cell, n_slots * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        HIP_OK(hipMemcpy(na.meta, buf[b].meta, n_slots * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        HIP_OK(hipMemcpy(na.rank, buf[b].rank, n_slots * sizeof(uint32_t), hipMemcpyDeviceToDevice));
      }
      free_arrays(buf[b]);
      buf[b] = na;
    }
    hipFree(pref);
    HIP_OK(hipMalloc(&pref, ncap * sizeof(float2)));
    HIP_OK(hipMemset(pref, 0, ncap * sizeof(float2)));
    hipFree(destroyed);
    destroyed_cap = (uint32_t)ncap;
    HIP_OK(hipMalloc(&destroyed, (uint64_t)destroyed_cap * sizeof(uint2)));
    hipFree(wp_events);
    wp_events_cap = (uint32_t)ncap;
    HIP_OK(hipMalloc(&wp_events, (uint64_t)wp_events_cap * sizeof(WpEvent)));
    hipFree(blk_desc);
    blk_desc_cap = (uint32_t)(ncap / 32 + ncells / std::max<uint64_t>(nx, 1) + 8);
    HIP_OK(hipMalloc(&blk_desc, (uint64_t)blk_desc_cap * sizeof(BlockDesc)));
    if (!n_blocks_dev) HIP_OK(hipMalloc(&n_blocks_dev, sizeof(uint32_t)));
    cap = ncap;
    return 0;
  }

  // ---- profiling helpers ----
  hipEvent_t get_event() {
    if (!event_pool.empty()) {
      hipEvent_t e = event_pool.back();
      event_pool.pop_back();
      return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
  }
  void prof_begin(uint32_t k) {
    prof_open = (profiling >> k) & 1u;
    if (!prof_open) return;
    Timed t;
    t.a = get_event();
    t.b = get_event();
    t.k = k;
    hipEventRecord(t.a, stream);
    timed.push_back(t);
  }
  void prof_end() {
    if (!prof_open) return;
    prof_open = false;
    hipEventRecord(timed.back().b, stream);
  }
  void prof_collect() {
    if (timed.empty()) return;
    hipStreamSynchronize(stream);
    for (auto& t : timed) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
        prof_ms[t.k] += ms;
        prof_n[t.k] += 1;
      }
      event_pool.push_back(t.a);
      event_pool.push_back(t.b);
    }
    timed.clear();
  }

  // ---- conversions ----
  // location_to_index (location_hash_2d.rs:54-66) in f64, then the offset from
  // the stored cell's geometric origin, rounded once to f32.
  // returns 0 = ok, 1 = "Index out of bounds", 2 = valid but not in this tile's owned cells
  int to_cell(double x, double y, uint32_t* cell, float* ox, float* oy) const {
    uint64_t xi = sat_usize((x - grid.offset_x) / grid.cell_size);
    uint64_t yi = sat_usize((y - grid.offset_y) / grid.cell_size);
    if (!tile) {
      unsigned __int128 idx = (unsigned __int128)xi * nx + yi;
      if (idx >= ncells) return 1;
      uint64_t flat = (uint64_t)idx;
      uint64_t sx = flat / nx, sy = flat % nx;
      *cell = (uint32_t)flat;
      *ox = (float)((x - grid.offset_x) - (double)sx * grid.cell_size);
      *oy = (float)((y - grid.offset_y) - (double)sy * grid.cell_size);
      return 0;
    }
    // tiles have neighbours instead of clamp / alias edges: stay inside the global grid
    if (!((x - grid.offset_x) >= 0.0) || !((y - grid.offset_y) >= 0.0) || xi >= gny || yi >= gnx)
      return 1;
    if (xi < gdev.org_x + gdev.own_x0 || xi >= gdev.org_x + gdev.own_x1 ||
        yi < gdev.org_y + gdev.own_y0 || yi >= gdev.org_y + gdev.own_y1)
      return 2;
    *cell = (uint32_t)((xi - gdev.org_x) * nx + (yi - gdev.org_y));
    *ox = (float)((x - grid.offset_x) - (double)xi * grid.cell_size);
    *oy = (float)((y - grid.offset_y) - (double)yi * grid.cell_size);
    return 0;
  }
  void to_global(uint32_t cell, float ox, float oy, double* x, double* y) const {
    uint64_t sx = cell / nx + gdev.org_x, sy = cell % nx + gdev.org_y;
    *x = grid.offset_x + ((double)sx * grid.cell_size + (double)ox);
    *y = grid.offset_y + ((double)sy * grid.cell_size + (double)oy);
  }

  // ---- tables ----
  uint32_t make_group(uint32_t hlp, uint32_t lp, double eyesight, int32_t sink) {
    for (uint32_t g = 0; g < groups.size(); ++g)
      if (groups[g].hlp == hlp && groups[g].lp == lp && groups[g].eyesight == eyesight &&
          groups[g].sink == sink)
        return g;
    groups.push_back(HostGroup{hlp, lp, eyesight, sink});
    groups_dirty = true;
    return (uint32_t)groups.size() - 1;
  }

  int upload_groups() {
    if (!groups_dirty) return 0;
    std::vector<GroupDev> g(groups.size());
    any_callback_hlp = false;
    max_eyesight = 0.0;
    for (size_t i = 0; i < groups.size(); ++i) {
      GroupDev& d = g[i];
      d.eyesight = (float)groups[i].eyesight;
      if (lp_kinds[groups[i].lp] == 1u) max_eyesight = std::max(max_eyesight, groups[i].eyesight);
      d.lp_kind = lp_kinds[groups[i].lp];
      const cs_zanlungo_params& z = lp_params[groups[i].lp];
      d.A = (float)z.agent_scale;
      d.D = (float)z.force_distance;
      d.inv_mass = 1.0f / (float)z.agent_mass;  // force * (1/m), zanlungo.rs:216
      d.R = (float)z.agent_radius;
      const cs_hlp_desc& h = hlps[groups[i].hlp];
      d.hlp_kind = h.kind;
      d.hvx = (float)h.vx;
      d.hvy = (float)h.vy;
      d.sink = groups[i].sink;
      if (d.sink >= 0 && !sinks[d.sink].alive) d.sink = -1;  // removed sink: no more waypoint tests
      if (h.kind == CS_HLP_CALLBACK) any_callback_hlp = true;
    }
    if (!groups_dev) HIP_OK(hipMalloc(&groups_dev, CS_MAX_GROUPS * sizeof(GroupDev)));
    if (!g.empty())
      HIP_OK(hipMemcpyAsync(groups_dev, g.data(), g.size() * sizeof(GroupDev), hipMemcpyHostToDevice,
                            stream));
    HIP_OK(hipStreamSynchronize(stream));
    groups_dirty = false;
    return 0;
  }

  int upload_sinks() {
    if (!sinks_dirty) return 0;
    HIP_OK(hipStreamSynchronize(stream));
    size_t ns = sinks.size();
    std::vector<SinkDev> sd(ns);
    std::vector<double> wps;
    std::vector<std::pair<uint32_t, uint32_t>> by_cell;  // (cell, slot)
    n_live_sinks = 0;
    for (size_t s = 0; s < ns; ++s) {
      const HostSink& h = sinks[s];
      SinkDev& d = sd[s];
      d.src_x = h.d.source_x;
      d.src_y = h.d.source_y;
      d.src_cell = CS_INVALID_CELL;
      d.src_ox = d.src_oy = 0;
      if (h.alive) {
        uint32_t c;
        float ox, oy;
        if (to_cell(h.d.source_x, h.d.source_y, &c, &ox, &oy) == 0) {
          d.src_cell = c;
          d.src_ox = ox;
          d.src_oy = oy;
          by_cell.push_back({c, (uint32_t)s});
        }
        ++n_live_sinks;
      }
      d.radius_sink = (float)h.d.radius_sink;
      d.wp_begin = (uint32_t)(wps.size() / 2);
      d.wp_count = (uint32_t)(h.waypoints.size() / 2);
      wps.insert(wps.end(), h.waypoints.begin(), h.waypoints.end());
      d.loop_forever = h.d.loop_forever ? 1u : 0u;
      d.group = h.group;
      d.eyesight = (float)h.d.agent_eyesight_range;
    }
    hipFree(sinks_dev); hipFree(waypoints_dev); hipFree(src_sorted); hipFree(src_occupied);
    hipFree(want_dev); hipFree(spawned_slots_dev); hipHostFree(want_host);
    want_host = nullptr;
    sinks_dev = nullptr; waypoints_dev = nullptr; src_sorted = nullptr; src_occupied = nullptr;
    want_dev = nullptr; spawned_slots_dev = nullptr;
    size_t nalloc = std::max<size_t>(ns, 1);
    HIP_OK(hipMalloc(&sinks_dev, nalloc * sizeof(SinkDev)));
    HIP_OK(hipMalloc(&waypoints_dev, std::max<size_t>(wps.size(), 2) * sizeof(double)));
    HIP_OK(hipMalloc(&src_sorted, nalloc * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&src_occupied, nalloc * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&want_dev, nalloc * sizeof(uint32_t)));
    HIP_OK(hipHostMalloc(&want_host, nalloc * sizeof(uint32_t)));
    HIP_OK(hipMalloc(&spawned_slots_dev, nalloc * sizeof(uint32_t)));
    HIP_OK(hipMemset(src_occupied, 0, nalloc * sizeof(uint32_t)));
    if (ns) HIP_OK(hipMemcpy(sinks_dev, sd.data(), ns * sizeof(SinkDev), hipMemcpyHostToDevice));
    if (!wps.empty())
      HIP_OK(hipMemcpy(waypoints_dev, wps.data(), wps.size() * sizeof(double), hipMemcpyHostToDevice));
    // static source grid: sources sorted by cell + per-cell start
    std::sort(by_cell.begin(), by_cell.end());
    std::vector<uint32_t> start(ncells + 1, 0), sorted_slots(by_cell.size());
    for (auto& bc : by_cell) start[bc.first + 1]++;
    for (uint64_t c = 0; c < ncells; ++c) start[c + 1] += start[c];
    for (size_t k = 0; k < by_cell.size(); ++k) sorted_slots[k] = by_cell[k].second;
    if (!src_cell_start) HIP_OK(hipMalloc(&src_cell_start, (ncells + 1) * sizeof(uint32_t)));
    HIP_OK(hipMemcpy(src_cell_start, start.data(), (ncells + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!sorted_slots.empty())
      HIP_OK(hipMemcpy(src_sorted, sorted_slots.data(), sorted_slots.size() * sizeof(uint32_t),
                       hipMemcpyHostToDevice));
    sinks_dirty = false;
    occ_valid = false;
    groups_dirty = true;  // a removed sink detaches its group
    return 0;
  }

  // ---- histogram of buf[cur] (only when the one kept by the step kernel is stale) ----
  int recount() {
    if (hist_valid) return 0;
    HIP_OK(hipMemsetAsync(cell_count, 0, (ncells + 1) * sizeof(uint32_t), stream));
    if (n_slots)
      hipLaunchKernelGGL(k_count, dim3((n_slots + 255) / 256), dim3(256), 0, stream, buf[cur], 0u,
                         n_slots, cell_count, ctr, gdev.tile);
    HIP_OK(hipGetLastError());
    hist_valid = true;
    return 0;
  }

  // ---- rebuild: histogram (if stale) -> scan -> scatter; leaves buf[cur] sorted ----
  int rebuild() {
    if (sorted) return 0;
    if (int rc = recount()) return rc;
    prof_begin(CS_K_SCAN);
    hipLaunchKernelGGL(k_scan_totals, dim3(n_scan_blocks), dim3(SCAN_BLOCK), 0, stream, cell_count,
                       (uint32_t)ncells, block_totals);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_scan_blocks), dim3(SCAN_BLOCK), 0, stream, cell_count,
                       (uint32_t)ncells, block_totals, n_scan_blocks, cell_start, ctr, n_blocks_dev);
    prof_end();
    prof_begin(CS_K_SCATTER);
    if (n_slots)
      hipLaunchKernelGGL(k_scatter, dim3((n_slots + 255) / 256), dim3(256), 0, stream, buf[cur],
                         buf[cur ^ 1], n_slots, cell_start, ctr, gdev.tile);
    prof_end();
    HIP_OK(hipGetLastError());
    cur ^= 1;
    sorted = true;
    hist_valid = false;  // the scan zeroed cell_count
    return 0;
  }

  int mark_occupancy() {
    if (occ_valid || n_live_sinks == 0) return 0;
    HIP_OK(hipMemsetAsync(src_occupied, 0, std::max<size_t>(sinks.size(), 1) * sizeof(uint32_t), stream));
    if (n_slots)
      hipLaunchKernelGGL(k_mark_sources, dim3((n_slots + 255) / 256), dim3(256), 0, stream, gdev,
                         buf[cur], n_slots, sinks_dev, src_cell_start, src_sorted, src_occupied, ctr);
    HIP_OK(hipGetLastError());
    occ_valid = true;
    return 0;
  }

  int read_counters(Counters* out) {
    HIP_OK(hipMemcpyAsync(ctr_host, ctr, sizeof(Counters), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    *out = *ctr_host;
    return 0;
  }

  // ---- host copy of buf[cur] ----
  struct HostState {
    std::vector<float2> off, vel;
    std::vector<uint32_t> id, cell, meta;
  };
  int download(HostState* h) {
    uint32_t n = n_slots;
    h->off.resize(n); h->vel.resize(n); h->id.resize(n); h->cell.resize(n); h->meta.resize(n);
    if (!n) return 0;
    const AgentArrays& a = buf[cur];
    HIP_OK(hipMemcpyAsync(h->off.data(), a.off, n * sizeof(float2), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(h->vel.data(), a.vel, n * sizeof(float2), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(h->id.data(), a.id, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(h->cell.data(), a.cell, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(h->meta.data(), a.meta, n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return 0;
  }

  // ---- callback high-level planners: the documented slow path ----
  // get_desired_velocity for every agent of a CALLBACK planner, batched per
  // planner, on the sorted state (lib.rs:264-273).
  int eval_callback_hlps() {
    HostState h;
    if (int rc = download(&h)) return rc;
    uint32_t n = n_slots;
    std::vector<float2> pv(n, make_float2(0.f, 0.f));
    for (uint32_t p = 0; p < hlps.size(); ++p) {
      if (hlps[p].kind != CS_HLP_CALLBACK || !hlps[p].velocity) continue;
      std::vector<uint32_t> slots;
      for (uint32_t i = 0; i < n; ++i)
        if (h.cell[i] != CS_INVALID_CELL && groups[h.meta[i] & 0xFFFFu].hlp == p) slots.push_back(i);
      if (slots.empty()) continue;
      size_t m = slots.size();
      std::vector<uint64_t> ids(m);
      std::vector<double> pos(2 * m), vel(2 * m), out(2 * m, 0.0);
      std::vector<uint8_t> some(m, 0);
      for (size_t k = 0; k < m; ++k) {
        uint32_t i = slots[k];
        ids[k] = h.id[i];
        to_global(h.cell[i], h.off[i].x, h.off[i].y, &pos[2 * k], &pos[2 * k + 1]);
        vel[2 * k] = h.vel[i].x;
        vel[2 * k + 1] = h.vel[i].y;
      }
      hlps[p].velocity(hlps[p].user, m, ids.data(), pos.data(), vel.data(), 0.0, out.data(), some.data());
      for (size_t k = 0; k < m; ++k)
        if (some[k]) pv[slots[k]] = make_float2((float)out[2 * k], (float)out[2 * k + 1]);
    }
    if (n) {
      HIP_OK(hipMemcpyAsync(pref, pv.data(), n * sizeof(float2), hipMemcpyHostToDevice, stream));
      HIP_OK(hipStreamSynchronize(stream));
    }
    return 0;
  }

  // ---- Simulation::add_agents, lib.rs:119-156 ----
  int add_agents(const double* xy, size_t n, uint32_t group, uint32_t owner, uint64_t* out_ids) {
    if (n == 0) return 0;
    if (next_id + n >= 0xFFFFFFFFull) {
      error = "agent id space exhausted (device ids are 32-bit)";
      return 4;
    }
    if (int rc = reserve((uint64_t)n_slots + n)) return rc;
    std::vector<float2> off, vel;
    std::vector<uint32_t> ids, cells, meta;
    off.reserve(n); ids.reserve(n); cells.reserve(n);
    int rc = 0;
    for (size_t k = 0; k < n; ++k) {
      uint32_t c;
      float ox, oy;
      uint64_t id = next_id++;  // consumed even when the insert fails (lib.rs:128-129)
      int where = to_cell(xy[2 * k], xy[2 * k + 1], &c, &ox, &oy);
      if (where == 1) {
        // The reference has already put the agent into `agents` when the index insert
        // fails (lib.rs:133-149); such an agent can never be stepped, so the engine
        // drops it and reports the same error.
        error = "Index out of bounds";
        rc = 1;
        break;
      }
      if (out_ids) out_ids[k] = id;
      if (where == 2) continue;  // another tile owns it
      off.push_back(make_float2(ox, oy));
      ids.push_back((uint32_t)id);
      cells.push_back(c);
      cs_event ev;
      ev.kind = CS_EVENT_SPAWNED;
      ev.source_sink = owner;
      ev.id = id;
      ev.x = xy[2 * k];
      ev.y = xy[2 * k + 1];
      if (record_events) events.push_back(ev);
    }
    const size_t ok = ids.size();
    vel.assign(ok, make_float2(0.f, 0.f));
    meta.assign(ok, group);
    if (ok) {
      AgentArrays& a = buf[cur];
      uint32_t at = n_slots;
      HIP_OK(hipMemcpyAsync(a.off + at, off.data(), ok * sizeof(float2), hipMemcpyHostToDevice, stream));
      HIP_OK(hipMemcpyAsync(a.vel + at, vel.data(), ok * sizeof(float2), hipMemcpyHostToDevice, stream));
      HIP_OK(hipMemcpyAsync(a.id + at, ids.data(), ok * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
      HIP_OK(hipMemcpyAsync(a.cell + at, cells.data(), ok * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
      HIP_OK(hipMemcpyAsync(a.meta + at, meta.data(), ok * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
      HIP_OK(hipStreamSynchronize(stream));  // the host vectors die at return
      n_slots += (uint32_t)ok;
      n_alive_host += ok;
      sorted = false;
      hist_valid = false;
      occ_valid = false;
      HIP_OK(hipMemcpy(&ctr->n_pending, &n_slots, sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return rc;
  }

  // CrowdGenerator::get_number_to_spawn for every live sink, ascending handle (lib.rs:199-222);
  // want[s] = 1 iff the generator asked for at least one agent.  Returns the number of 1s.
  uint32_t eval_generators(double dt_seconds, uint32_t* want) {
    uint32_t n_want = 0;
    for (size_t s = 0; s < sinks.size(); ++s) {
      HostSink& h = sinks[s];
      want[s] = 0;
      if (!h.alive) continue;
      uint64_t call = h.calls++;
      uint64_t nsp = 0;
      switch (h.d.generator_kind) {
        case CS_GEN_MONOTONIC: {  // (dt * rate).round() as usize, source_sink.rs:96-100
          double v = std::round(dt_seconds * h.d.rate);
          nsp = v > 0 ? (uint64_t)v : 0;
          break;
        }
        case CS_GEN_POISSON_SEEDED:
          nsp = poisson_seeded(h.d.seed, call, dt_seconds * h.d.rate);
          break;
        case CS_GEN_CALLBACK:
          nsp = h.d.generator ? (uint64_t)h.d.generator(h.d.generator_user, dt_seconds) : 0;
          break;
      }
      want[s] = nsp > 0 ? 1u : 0u;  // the loop over n is commented out (lib.rs:207)
      n_want += want[s];
    }
    return n_want;
  }

  // ---- tile engines: Phase A split in two so that ids follow the GLOBAL sink order ----
  // probe: flags[s] = 1 iff sink s is owned by this tile, its generator fired and nobody (owned
  // agent or ghost) stands within 0.4 of its source.
  int spawn_probe(double dt_seconds, uint8_t* flags) {
    if (int rc = upload_sinks()) return rc;
    if (int rc = upload_groups()) return rc;
    std::memset(flags, 0, sinks.size());
    uint32_t n_want = eval_generators(dt_seconds, want_host);
    if (!n_want) return 0;
    occ_valid = false;  // ghosts arrived since the step kernel marked its own agents
    if (int rc = mark_occupancy()) return rc;
    std::vector<uint32_t> occ(sinks.size());
    HIP_OK(hipMemcpyAsync(occ.data(), src_occupied, sinks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost,
                          stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<SinkDev> dummy;
    for (size_t s = 0; s < sinks.size(); ++s) {
      if (!want_host[s] || occ[s]) continue;
      uint32_t c;
      float ox, oy;
      if (to_cell(sinks[s].d.source_x, sinks[s].d.source_y, &c, &ox, &oy) == 0) flags[s] = 1;
    }
    return 0;
  }
  // commit: `flags` is the OR over all tiles.  Ids are next_id + rank in ascending handle order;
  // this tile appends the agents of the sinks it owns.
  int spawn_commit(const uint8_t* flags) {
    std::vector<SpawnRecord> mine;
    uint64_t id = next_id;
    for (size_t s = 0; s < sinks.size(); ++s) {
      if (!flags[s]) continue;
      const uint64_t my_id = id++;
      uint32_t c;
      float ox, oy;
      if (to_cell(sinks[s].d.source_x, sinks[s].d.source_y, &c, &ox, &oy) != 0) continue;
      if (my_id >= 0xFFFFFFFFull) {
        error = "agent id space exhausted (device ids are 32-bit)";
        return 4;
      }
      mine.push_back(SpawnRecord{ox, oy, (uint32_t)my_id, c, sinks[s].group, 0u});
      if (record_events) {
        cs_event ev;
        ev.kind = CS_EVENT_SPAWNED;
        ev.source_sink = (uint32_t)s;
        ev.id = my_id;
        ev.x = sinks[s].d.source_x;
        ev.y = sinks[s].d.source_y;
        events.push_back(ev);
      }
      const cs_hlp_desc& p = hlps[sinks[s].d.hlp];
      if (p.kind == CS_HLP_CALLBACK && p.set_target && !sinks[s].waypoints.empty())
        p.set_target(p.user, my_id, sinks[s].d.source_x, sinks[s].d.source_y, sinks[s].waypoints[0],
                     sinks[s].waypoints[1], sinks[s].d.radius_sink, sinks[s].d.radius_sink);
    }
    next_id = id;
    committed_spawns = (uint32_t)mine.size();
    spawn_committed = true;
    if (mine.empty()) return 0;
    if (int rc = recount()) return rc;
    if (int rc = reserve((uint64_t)n_slots + mine.size())) return rc;
    if (mine.size() > spawn_rec_cap) {
      hipFree(spawn_rec_dev);
      spawn_rec_cap = (uint32_t)mine.size() * 2u + 64u;
      HIP_OK(hipMalloc(&spawn_rec_dev, (size_t)spawn_rec_cap * sizeof(SpawnRecord)));
    }
    HIP_OK(hipMemcpyAsync(spawn_rec_dev, mine.data(), mine.size() * sizeof(SpawnRecord),
                          hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_append_spawns, dim3(((uint32_t)mine.size() + 255) / 256), dim3(256), 0, stream,
                       buf[cur], (uint32_t)cap, spawn_rec_dev, (uint32_t)mine.size(), cell_count, ctr);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(stream));  // `mine` dies at return
    n_slots += (uint32_t)mine.size();
    sorted = false;
    return 0;
  }

  // ---- Simulation::step, lib.rs:195-383 ----
  int step(double dt_seconds, cs_step_report* report) {
    if (poisoned) {
      error = "Index out of bounds";
      return 1;
    }
    if (int rc = upload_sinks()) return rc;
    if (int rc = upload_groups()) return rc;
    const bool has_sinks = n_live_sinks > 0;
    if (tile && has_sinks && !spawn_committed) {
      error = "tile engine with source-sinks: run cs_spawn_probe / cs_spawn_commit before cs_step";
      return 6;
    }
    // a tile's slot count changes with every halo exchange: the host re-reads it each step
    const bool need_host = has_sinks || any_callback_hlp || report != nullptr || tile;

    if (has_sinks)  // n_spawned (n_halo_overflow is sticky: it poisons the tile)
      HIP_OK(hipMemsetAsync(&ctr->n_spawned, 0, sizeof(uint32_t), stream));

    // ---- Phase A: spawn (lib.rs:199-254) ----
    uint32_t n_want = 0;
    bool spawn_events_done = false;
    const uint64_t first_spawn_id = next_id;
    if (has_sinks && !tile) {
      uint32_t* want = want_host;  // the previous step ended with a sync: its copy is done
      n_want = eval_generators(dt_seconds, want);
      if (n_want) {
        if (next_id + n_want >= 0xFFFFFFFFull) {
          error = "agent id space exhausted (device ids are 32-bit)";
          return 4;
        }
        if (int rc = reserve((uint64_t)n_slots + n_want)) return rc;
        if (int rc = mark_occupancy()) return rc;  // no-op right after a step
        if (int rc = recount()) return rc;         // no-op right after a step
        HIP_OK(hipMemcpyAsync(want_dev, want, sinks.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                              stream));
        prof_begin(CS_K_SPAWN);
        hipLaunchKernelGGL(k_spawn, dim3(1), dim3(1024), 0, stream, buf[cur], n_slots, (uint32_t)cap,
                           sinks_dev, want_dev, (uint32_t)sinks.size(), n_want, src_occupied,
                           cell_count, (uint32_t)first_spawn_id, spawned_slots_dev, ctr);
        prof_end();
        HIP_OK(hipGetLastError());
        n_slots += n_want;
        sorted = false;
        if (any_callback_hlp) {
          // a host planner must see set_target(new agent) BEFORE this step asks it for a velocity
          // (lib.rs:236-250 runs before the update loop): read the spawn result back now
          Counters c0;
          if (int rc = read_counters(&c0)) return rc;
          if (int rc = finish_spawn_events(c0.n_spawned, first_spawn_id)) return rc;
          spawn_events_done = true;
        }
      }
    }

    // ---- index for this step (location_hash_2d.rs:126-149) ----
    const bool rebuilt = !sorted;
    if (!rebuilt)  // no re-sort this step: the scan is what normally zeroes the step counters
      HIP_OK(hipMemsetAsync(&ctr->n_destroyed, 0, 6 * sizeof(uint32_t), stream));
    if (int rc = rebuild()) return rc;

    // ---- HighLevelPlanner callbacks (slow path) ----
    if (any_callback_hlp)
      if (int rc = eval_callback_hlps()) return rc;

    // ---- Phases B + C: per-agent update and commit (lib.rs:259-359) ----
    if (has_sinks)
      HIP_OK(hipMemsetAsync(src_occupied, 0, std::max<size_t>(sinks.size(), 1) * sizeof(uint32_t), stream));
    // cell_count is all zero here: the scan clears it while reading (k_scan_apply)
    if (tile && n_slots)  // ghosts get no thread: their output slots must read "dead"
      HIP_OK(hipMemsetAsync(buf[cur ^ 1].cell, 0xFF, (size_t)n_slots * sizeof(uint32_t), stream));
    StepParams P;
    P.g = gdev;
    P.dt = (float)dt_seconds;
    P.n = n_slots;
    P.has_sinks = has_sinks ? 1u : 0u;
    P.n_src_cells = 0;
    EpilogueCtx E;
    E.out = buf[cur ^ 1];
    E.cell_count = cell_count;
    E.ctr = ctr;
    E.groups = groups_dev;
    E.sinks = sinks_dev;
    E.waypoints = waypoints_dev;
    E.grid_off_x = grid.offset_x;
    E.grid_off_y = grid.offset_y;
    E.cell_size = grid.cell_size;
    E.destroyed = destroyed;
    E.destroyed_cap = destroyed_cap;
    E.wp_events = wp_events;
    E.wp_events_cap = wp_events_cap;
    E.src_cell_start = src_cell_start;
    E.src_sorted = src_sorted;
    E.src_occupied = src_occupied;
    // neighbour kernel: LDS-tiled strips when the crowd is large enough to fill them
    const uint64_t n_rows = ncells / std::max<uint64_t>(nx, 1);
    int h = max_eyesight > 0.0 ? (int)std::ceil(max_eyesight / grid.cell_size - 1e-6) : 0;
    bool tiled = n_slots >= 2048 && h >= 1 && 2 * h + 1 <= TILE_MAX_ROWS && nx >= (uint64_t)(3 * h + 2);
    if (flags & CS_CFG_FORCE_GATHER) tiled = false;
    if ((flags & CS_CFG_FORCE_TILED) && h >= 1 && 2 * h + 1 <= TILE_MAX_ROWS) tiled = true;
    if (tiled) {
      TileCfg cfg;
      cfg.h = h;
      cfg.debug = (flags >> 8) & 0xFFu;
      // LDS budget per workgroup: staged agents (20 B each, ~(2h+1) strips of 256 + halo ends),
      // the cell table and the per-thread neighbour lists; sized so that `tile_blocks_per_cu`
      // workgroups fit in the 160 KiB of a CU
      const uint32_t rb = tile_rows;
      if (rb <= 1)
        cfg.agents_cap = ((uint32_t)(2 * h + 1) * TILE_THREADS * 9u / 8u + 192u + tile_agents_slack + 63u) & ~63u;
      else
        cfg.agents_cap = ((rb + 2u * h) * (tile_target / rb) * 5u / 4u + 192u + tile_agents_slack + 63u) & ~63u;
      cfg.agents_cap = std::min<uint32_t>(6144u, cfg.agents_cap);
      cfg.table_cap = 1024u * (uint32_t)(h > 1 ? 2 : 1);
      const bool e16 = h <= 1 && cfg.agents_cap <= 4096u;
      const size_t entry = e16 ? 2u : 4u;
      const size_t fixed = (size_t)cfg.agents_cap * 20u + (size_t)cfg.table_cap * 2u + 256u;
      const size_t budget = (size_t)(160u * 1024u) / std::max(1u, tile_blocks_per_cu);
      cfg.list_cap = tile_list_cap;
      if (!cfg.list_cap) {
        cfg.list_cap = 16u;
        while (cfg.list_cap + 8u <= 64u && fixed + (size_t)(cfg.list_cap + 8u) * TILE_THREADS * entry <= budget)
          cfg.list_cap += 8u;
      }
      cfg.list_cap = std::min(cfg.list_cap, 64u);
      while ((size_t)cfg.list_cap * TILE_THREADS * entry < (size_t)cfg.agents_cap * 4u) cfg.list_cap += 8u;
      size_t lds = (size_t)cfg.agents_cap * 20u + (size_t)cfg.list_cap * TILE_THREADS * entry +
                   (size_t)cfg.table_cap * 2u;
      uint32_t grid_blocks;
      if (rb <= 1) {
        hipLaunchKernelGGL(k_build_blocks, dim3(1), dim3(1024), 0, stream, gdev, cell_start, blk_desc,
                           blk_desc_cap, n_blocks_dev, ctr);
        grid_blocks = (n_slots + TILE_THREADS - 1u) / TILE_THREADS + (uint32_t)std::min<uint64_t>(n_rows, n_slots);
      } else {
        const uint32_t own_rows = gdev.own_x1 - gdev.own_x0;
        const uint32_t n_bands = (own_rows + rb - 1u) / rb;
        if (!rebuilt) HIP_OK(hipMemsetAsync(n_blocks_dev, 0, sizeof(uint32_t), stream));  // else the scan did
        hipLaunchKernelGGL(k_build_bands, dim3(n_bands), dim3(256), 0, stream, gdev, cell_start, rb,
                           tile_target, band_prefix, blk_desc, blk_desc_cap, n_blocks_dev, ctr);
        grid_blocks = n_slots / tile_target + n_bands + 1u;
        grid_blocks = std::max(grid_blocks, (n_slots + TILE_THREADS - 1u) / TILE_THREADS);
      }
      prof_begin(CS_K_NEIGHBOUR_FORCE);
      if (n_slots)
      {
        if (e16)
          hipLaunchKernelGGL(k_step_tiled<true>, dim3(grid_blocks), dim3(TILE_THREADS), lds, stream, P,
                             buf[cur], E, cell_start, pref, blk_desc, n_blocks_dev, cfg);
        else
          hipLaunchKernelGGL(k_step_tiled<false>, dim3(grid_blocks), dim3(TILE_THREADS), lds, stream, P,
                             buf[cur], E, cell_start, pref, blk_desc, n_blocks_dev, cfg);
      }
      prof_end();
    } else {
      if (tile)  // owned count + slot count of the step output
        hipLaunchKernelGGL(k_build_blocks, dim3(1), dim3(1024), 0, stream, gdev, cell_start, blk_desc,
                           blk_desc_cap, n_blocks_dev, ctr);
      prof_begin(CS_K_NEIGHBOUR_FORCE);
      if (n_slots)
        hipLaunchKernelGGL(k_step_gather, dim3((n_slots + 255) / 256), dim3(256), 0, stream, P, buf[cur],
                           E, cell_start, pref);
      prof_end();
    }
    HIP_OK(hipGetLastError());

    if (!need_host) {
      // fire and forget: an out-of-bounds agent poisons the engine at the next sync
      cur ^= 1;
      sorted = false;
      hist_valid = true;
      occ_valid = has_sinks;
      return 0;
    }

    Counters c;
    if (int rc = read_counters(&c)) return rc;
    if (c.n_out_of_bounds) {
      // "Index out of bounds" (location_hash_2d.rs:61-63 via lib.rs:299-302): nothing is
      // committed; the pre-step state (including this step's spawns) stays current.
      HIP_OK(hipMemsetAsync(&ctr->n_out_of_bounds, 0, sizeof(uint32_t), stream));
      HIP_OK(hipMemsetAsync(cell_count, 0, (ncells + 1) * sizeof(uint32_t), stream));  // partial histogram
      hist_valid = false;
      occ_valid = false;
      n_alive_host = c.n_alive;
      n_slots = c.n_alive;
      if (!spawn_events_done) finish_spawn_events(c.n_spawned, first_spawn_id);
      error = "Index out of bounds";
      return 1;
    }
    cur ^= 1;
    sorted = false;
    hist_valid = true;
    occ_valid = has_sinks;
    n_slots = c.n_alive;  // the step wrote exactly the live population
    if (!spawn_events_done)
      if (int rc = finish_spawn_events(c.n_spawned, first_spawn_id)) return rc;
    if (int rc = finish_destroy_events(c)) return rc;
    n_alive_host = tile ? (uint64_t)c.n_owned - c.n_destroyed : (uint64_t)c.n_alive - c.n_destroyed;
    if (c.n_halo_overflow) {
      error = "halo buffer overflow: raise capacity_records of cs_halo_set_buffers";
      poisoned = true;
      return 7;
    }
    if (report) {
      report->n_agents = n_alive_host;
      report->n_spawned = tile ? committed_spawns : c.n_spawned;
      report->n_destroyed = c.n_destroyed;
      report->n_waypoint_hits = c.n_waypoint_hits;
      report->n_tti_zero = c.n_tti_zero;
      report->n_nonfinite = c.n_nonfinite;
      report->n_clamped = c.n_clamped;
    }
    spawn_committed = false;
    committed_spawns = 0;
    return 0;
  }

  // SPAWNED events + set_target for this step's spawns (lib.rs:151-153,236-250),
  // in ascending sink order.
  int finish_spawn_events(uint32_t n_spawned, uint64_t first_id) {
    if (!n_spawned) return 0;
    if (!record_events && !any_callback_hlp) {  // nobody listens: ids advance, nothing to read back
      next_id = first_id + n_spawned;
      return 0;
    }
    std::vector<uint32_t> slots(n_spawned);
    HIP_OK(hipMemcpy(slots.data(), spawned_slots_dev, n_spawned * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < n_spawned; ++k) {
      const HostSink& h = sinks[slots[k]];
      cs_event ev;
      ev.kind = CS_EVENT_SPAWNED;
      ev.source_sink = slots[k];
      ev.id = first_id + k;
      ev.x = h.d.source_x;
      ev.y = h.d.source_y;
      if (record_events) events.push_back(ev);
      const cs_hlp_desc& p = hlps[h.d.hlp];
      if (p.kind == CS_HLP_CALLBACK && p.set_target && !h.waypoints.empty())
        p.set_target(p.user, ev.id, ev.x, ev.y, h.waypoints[0], h.waypoints[1], h.d.radius_sink,
                     h.d.radius_sink);
    }
    next_id = first_id + n_spawned;
    return 0;
  }

  // waypoint set_target callbacks (lib.rs:325-333) and DESTROYED events in
  // ascending id (lib.rs:378-380, canonical order).
  int finish_destroy_events(const Counters& c) {
    if (c.n_wp_events) {
      uint32_t m = std::min(c.n_wp_events, wp_events_cap);
      std::vector<WpEvent> w(m);
      HIP_OK(hipMemcpy(w.data(), wp_events, m * sizeof(WpEvent), hipMemcpyDeviceToHost));
      std::sort(w.begin(), w.end(), [](const WpEvent& a, const WpEvent& b) { return a.id < b.id; });
      for (const WpEvent& ev : w) {  // ascending id = the canonical visiting order
        const HostGroup& g = groups[ev.group];
        const cs_hlp_desc& p = hlps[g.hlp];
        if (p.kind != CS_HLP_CALLBACK || !p.set_target || g.sink < 0) continue;
        const HostSink& h = sinks[g.sink];
        if (2 * (size_t)ev.next_wp + 1 >= h.waypoints.size()) continue;
        double px, py;
        to_global(ev.cell, ev.ox, ev.oy, &px, &py);
        p.set_target(p.user, ev.id, px, py, h.waypoints[2 * ev.next_wp], h.waypoints[2 * ev.next_wp + 1],
                     h.d.radius_sink, h.d.radius_sink);
      }
    }
    if (c.n_destroyed && (record_events || any_callback_hlp)) {
      uint32_t m = std::min(c.n_destroyed, destroyed_cap);
      std::vector<uint2> d(m);
      HIP_OK(hipMemcpy(d.data(), destroyed, m * sizeof(uint2), hipMemcpyDeviceToHost));
      std::sort(d.begin(), d.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
      for (auto& it : d) {
        const HostGroup& g = groups[it.y & 0xFFFFu];
        const cs_hlp_desc& p = hlps[g.hlp];
        if (p.kind == CS_HLP_CALLBACK && p.remove_agent) p.remove_agent(p.user, it.x);
        cs_event ev;
        ev.kind = CS_EVENT_DESTROYED;
        ev.source_sink = g.sink >= 0 ? (uint32_t)g.sink : UINT32_MAX;
        ev.id = it.x;
        ev.x = ev.y = 0;
        if (record_events) events.push_back(ev);
      }
    }
    return 0;
  }

  // ---- tiles: pack / unpack one axis of the halo exchange ----
  int halo_pack(uint32_t axis) {
    if (poisoned) {
      error = "Index out of bounds";
      return 1;
    }
    if (int rc = recount()) return rc;  // ranks of appended records extend this histogram
    HaloDir& lo = halo[axis * 2];
    HaloDir& hi = halo[axis * 2 + 1];
    if (lo.send) HIP_OK(hipMemsetAsync(lo.send, 0, sizeof(HaloRecord), stream));
    if (hi.send) HIP_OK(hipMemsetAsync(hi.send, 0, sizeof(HaloRecord), stream));
    if ((lo.send || hi.send) && n_slots) {
      prof_begin(CS_K_HALO);
      hipLaunchKernelGGL(k_halo_pack, dim3((n_slots + 255) / 256), dim3(256), 0, stream, gdev, buf[cur],
                         n_slots, axis, 2u * halo_cells, lo.send, hi.send,
                         std::max(lo.cap, hi.cap), ctr);
      prof_end();
    }
    HIP_OK(hipGetLastError());
    return 0;
  }
  int halo_unpack(uint32_t axis) {
    for (int k = 0; k < 2; ++k) {
      HaloDir& d = halo[axis * 2 + k];
      if (!d.recv) continue;
      if ((uint64_t)n_slots + d.cap > cap)  // the tile's population grew: reallocate (rare)
        if (int rc = reserve((uint64_t)n_slots + d.cap)) return rc;
      prof_begin(CS_K_HALO);
      hipLaunchKernelGGL(k_halo_unpack, dim3((d.cap + 255) / 256), dim3(256), 0, stream, gdev, buf[cur],
                         (uint32_t)cap, d.recv, d.cap, cell_count, ctr);
      prof_end();
      n_slots += d.cap;  // upper bound; kernels stop at the device-side count
      sorted = false;
      occ_valid = false;
    }
    HIP_OK(hipGetLastError());
    return 0;
  }

  // make buf[cur] sorted for queries between steps
  int ensure_index() {
    if (sorted) return 0;
    return rebuild();
  }
};

extern "C" {

uint32_t cs_abi_version(void) { return CS_ABI_VERSION; }

void cs_destroy(cs_engine* e) {
  if (!e) return;
  hipSetDevice(e->device);
  if (e->stream) hipStreamSynchronize(e->stream);
  e->free_arrays(e->buf[0]);
  e->free_arrays(e->buf[1]);
  hipFree(e->pref); hipFree(e->cell_count); hipFree(e->cell_start); hipFree(e->block_totals);
  hipFree(e->ctr); hipHostFree(e->ctr_host); hipFree(e->destroyed); hipFree(e->wp_events);
  hipFree(e->groups_dev); hipFree(e->sinks_dev); hipFree(e->waypoints_dev);
  hipFree(e->src_cell_start); hipFree(e->src_sorted); hipFree(e->src_occupied);
  hipFree(e->want_dev); hipFree(e->spawned_slots_dev); hipHostFree(e->want_host); hipFree(e->blk_desc); hipFree(e->n_blocks_dev); hipFree(e->band_prefix); hipFree(e->spawn_rec_dev);
  for (auto& t : e->timed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
  for (auto ev : e->event_pool) hipEventDestroy(ev);
  if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
  delete e;
}

// Simulation::new(LocationHash2D::new(..)), lib.rs:103 + location_hash_2d.rs:33-51
cs_engine* cs_create(const cs_grid_desc* grid, const cs_device_cfg* cfg) {
  if (!grid) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    fprintf(stderr, "crowdstep: no HIP device visible; the engine has no CPU fallback\n");
    return nullptr;
  }
  cs_engine* e = new cs_engine();
  e->grid = *grid;
  e->device = cfg ? cfg->device_ordinal : 0;
  e->flags = cfg ? cfg->flags : 0;
  if (e->device < 0 || e->device >= ndev || hipSetDevice(e->device) != hipSuccess) {
    delete e;
    return nullptr;
  }
  // (width / cell) as usize is the row stride, used on BOTH axes, and the number of x rows
  // that fit is len / stride = (height / cell) as usize (location_hash_2d.rs:36-37,59)
  e->gnx = sat_usize(grid->width / grid->cell_size);
  e->gny = sat_usize(grid->height / grid->cell_size);
  unsigned __int128 gnc = (unsigned __int128)e->gnx * e->gny;
  if (gnc >= 0x7FFFFFFFull || e->gnx >= 0x7FFFFFFFull) {
    fprintf(stderr, "crowdstep: grid too large for 32-bit cell indices\n");
    delete e;
    return nullptr;
  }
  e->nx = e->gnx;
  e->ny = e->gny;
  std::memset(&e->gdev, 0, sizeof e->gdev);
  uint32_t ox0 = 0, ox1 = (uint32_t)e->gny, oy0 = 0, oy1 = (uint32_t)e->gnx, org_x = 0, org_y = 0;
  if (cfg && (cfg->tile_cx1 | cfg->tile_cy1)) {
    const uint32_t H = cfg->halo_cells;
    if (cfg->tile_cx0 >= cfg->tile_cx1 || cfg->tile_cy0 >= cfg->tile_cy1 || cfg->tile_cx1 > e->gny ||
        cfg->tile_cy1 > e->gnx || H == 0) {
      fprintf(stderr, "crowdstep: bad tile rectangle / halo_cells\n");
      delete e;
      return nullptr;
    }
    e->tile = true;
    e->halo_cells = H;
    const uint32_t lx0 = cfg->tile_cx0 > H ? cfg->tile_cx0 - H : 0;
    const uint32_t ly0 = cfg->tile_cy0 > H ? cfg->tile_cy0 - H : 0;
    const uint32_t lx1 = (uint32_t)std::min<uint64_t>(e->gny, (uint64_t)cfg->tile_cx1 + H);
    const uint32_t ly1 = (uint32_t)std::min<uint64_t>(e->gnx, (uint64_t)cfg->tile_cy1 + H);
    e->ny = lx1 - lx0;
    e->nx = ly1 - ly0;
    org_x = lx0;
    org_y = ly0;
    ox0 = cfg->tile_cx0 - lx0;
    ox1 = cfg->tile_cx1 - lx0;
    oy0 = cfg->tile_cy0 - ly0;
    oy1 = cfg->tile_cy1 - ly0;
  }
  e->ncells = e->nx * e->ny;
  e->gdev.nx = (uint32_t)e->nx;
  e->gdev.ny = (uint32_t)e->ny;
  e->gdev.ncells = (uint32_t)e->ncells;
  e->gdev.cs = (float)grid->cell_size;
  e->gdev.cs_lo = (float)(grid->cell_size - (double)e->gdev.cs);
  e->gdev.inv_cs = 1.0f / e->gdev.cs;
  e->gdev.tile = e->tile ? 1u : 0u;
  e->gdev.own_x0 = ox0;
  e->gdev.own_x1 = ox1;
  e->gdev.own_y0 = oy0;
  e->gdev.own_y1 = oy1;
  e->gdev.org_x = org_x;
  e->gdev.org_y = org_y;
  if (cfg && cfg->stream) {
    e->stream = (hipStream_t)cfg->stream;
  } else {
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
      delete e;
      return nullptr;
    }
    e->own_stream = true;
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, e->device);
  e->backend = std::string("hip:") + prop.gcnArchName;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_step_tiled<false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess)
    (void)hipGetLastError();  // not fatal: the default 64 KiB covers eyesight <= 2 cells
  if (const char* v = getenv("CS_TILE_BLOCKS_PER_CU")) e->tile_blocks_per_cu = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_TILE_LIST_CAP")) e->tile_list_cap = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_TILE_AGENTS_SLACK")) e->tile_agents_slack = (uint32_t)atoi(v);
  if (const char* v = getenv("CS_TILE_ROWS")) e->tile_rows = std::min<uint32_t>(TILE_MAX_OWN_ROWS, std::max(1, atoi(v)));
  if (const char* v = getenv("CS_TILE_TARGET")) e->tile_target = std::min<uint32_t>(TILE_THREADS, std::max(32, atoi(v)));
  bool ok = true;
  ok = ok && hipMalloc(&e->cell_count, (e->ncells + 1) * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc(&e->cell_start, (e->ncells + 1) * sizeof(uint32_t)) == hipSuccess;
  e->n_scan_blocks = (uint32_t)std::max<uint64_t>(1, (e->ncells + SCAN_TILE - 1) / SCAN_TILE);
  ok = ok && hipMalloc(&e->block_totals, e->n_scan_blocks * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc(&e->band_prefix, (e->ncells + 1) * sizeof(uint32_t)) == hipSuccess;
  ok = ok && hipMalloc(&e->ctr, sizeof(Counters)) == hipSuccess;
  ok = ok && hipHostMalloc(&e->ctr_host, sizeof(Counters)) == hipSuccess;
  if (ok) {
    hipMemset(e->cell_count, 0, (e->ncells + 1) * sizeof(uint32_t));
    hipMemset(e->cell_start, 0, (e->ncells + 1) * sizeof(uint32_t));
    hipMemset(e->ctr, 0, sizeof(Counters));
  }
  if (!ok || e->upload_sinks() != 0 ||
      e->reserve(cfg && cfg->capacity_hint ? cfg->capacity_hint : 1024) != 0) {
    fprintf(stderr, "crowdstep: device allocation failed: %s\n", e->error.c_str());
    cs_destroy(e);
    return nullptr;
  }
  return e;
}

const char* cs_last_error(const cs_engine* e) { return e ? e->error.c_str() : "null engine"; }
const char* cs_backend_name(const cs_engine* e) { return e ? e->backend.c_str() : ""; }

uint32_t cs_register_zanlungo(cs_engine* e, const cs_zanlungo_params* p) {
  e->lp_params.push_back(*p);
  e->lp_kinds.push_back(1u);
  return (uint32_t)e->lp_kinds.size() - 1;
}
uint32_t cs_register_no_local_plan(cs_engine* e) {
  cs_zanlungo_params z;
  std::memset(&z, 0, sizeof z);
  z.agent_mass = 1.0;
  z.force_distance = 1.0;
  e->lp_params.push_back(z);
  e->lp_kinds.push_back(0u);
  return (uint32_t)e->lp_kinds.size() - 1;
}
uint32_t cs_register_hlp(cs_engine* e, const cs_hlp_desc* d) {
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
  if (e->groups.size() + 1 >= CS_MAX_GROUPS) {
    e->error = "too many distinct (planner, eyesight) groups";
    return 5;
  }
  uint32_t g = e->make_group(hlp, lp, eyesight, -1);
  return e->add_agents(xy, n, g, UINT32_MAX, out_ids);
}

// Simulation::remove_agents, lib.rs:176-192
int cs_remove_agent(cs_engine* e, uint64_t id) {
  hipSetDevice(e->device);
  cs_engine::HostState h;
  if (int rc = e->download(&h)) return rc;
  for (uint32_t i = 0; i < e->n_slots; ++i) {
    if (h.cell[i] == CS_INVALID_CELL || h.id[i] != id) continue;
    uint32_t inv = CS_INVALID_CELL;
    if (hipMemcpy(e->buf[e->cur].cell + i, &inv, sizeof inv, hipMemcpyHostToDevice) != hipSuccess) {
      e->error = "HIP error while removing an agent";
      return 90;
    }
    const HostGroup& g = e->groups[h.meta[i] & 0xFFFFu];
    const cs_hlp_desc& p = e->hlps[g.hlp];
    if (p.kind == CS_HLP_CALLBACK && p.remove_agent) p.remove_agent(p.user, id);
    e->sorted = false;
    e->hist_valid = false;
    e->occ_valid = false;
    e->n_alive_host -= 1;
    cs_event ev;
    ev.kind = CS_EVENT_DESTROYED;
    ev.source_sink = g.sink >= 0 ? (uint32_t)g.sink : UINT32_MAX;
    ev.id = id;
    ev.x = ev.y = 0;
    if (e->record_events) e->events.push_back(ev);
    return 0;
  }
  e->error = "unknown agent id";
  return 2;
}

uint32_t cs_add_source_sink(cs_engine* e, const cs_source_sink_desc* d) {
  if (e->groups.size() + 1 >= CS_MAX_GROUPS || d->n_waypoints == 0 || d->n_waypoints > 65535) {
    e->error = "too many source-sinks / planner groups (65535) or bad waypoint count";
    return UINT32_MAX;
  }
  HostSink s;
  s.d = *d;
  s.waypoints.assign(d->waypoints_xy, d->waypoints_xy + 2 * d->n_waypoints);
  s.d.waypoints_xy = nullptr;
  uint32_t handle = (uint32_t)e->sinks.size();
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
  return e->step(dt_seconds, report);
}

int cs_synchronize(cs_engine* e) {
  hipSetDevice(e->device);
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

size_t cs_agent_count(cs_engine* e) { return (size_t)e->n_alive_host; }

size_t cs_read_agents(cs_engine* e, cs_agent_view* out, size_t cap) {
  hipSetDevice(e->device);
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
    out[k].next_waypoint = h.meta[i] >> 16;
    out[k].eyesight_range = e->groups[h.meta[i] & 0xFFFFu].eyesight;
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
  e->record_events = on != 0;
  if (!on) e->events.clear();
}

// SpatialIndex::get_neighbours_in_radius, location_hash_2d.rs:240-258
size_t cs_query_radius(cs_engine* e, double radius, double x, double y, uint64_t* out_ids, size_t cap) {
  hipSetDevice(e->device);
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
  hx = std::min(hx, (long long)e->nx); hy = std::min(hy, (long long)e->nx * 2);
  // query point relative to a reference cell: the cell of the point clamped into the grid
  long long qx = std::min(std::max(fl(x, e->grid.offset_x), 0ll), (long long)e->nx - 1);
  long long qy = std::min(std::max(fl(y, e->grid.offset_y), 0ll), (long long)e->nx - 1);
  float qox = (float)((x - e->grid.offset_x) - (double)qx * e->grid.cell_size);
  float qoy = (float)((y - e->grid.offset_y) - (double)qy * e->grid.cell_size);
  uint32_t qcap = (uint32_t)std::min<size_t>(cap, 1u << 20);
  uint32_t* d_out = nullptr;
  uint32_t* d_cnt = nullptr;
  if (hipMalloc(&d_out, std::max<uint32_t>(qcap, 1) * sizeof(uint32_t)) != hipSuccess) return 0;
  if (hipMalloc(&d_cnt, sizeof(uint32_t)) != hipSuccess) {
    hipFree(d_out);
    return 0;
  }
  hipLaunchKernelGGL(k_query_radius, dim3(1), dim3(64), 0, e->stream, e->gdev, e->buf[e->cur],
                     e->cell_start, lx, hx, ly, hy, (uint32_t)qx, (uint32_t)qy, qox, qoy, (float)radius,
                     d_out, qcap, d_cnt);
  uint32_t cnt = 0;
  hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, e->stream);
  hipStreamSynchronize(e->stream);
  uint32_t m = std::min(cnt, qcap);
  std::vector<uint32_t> ids(m);
  if (m) hipMemcpy(ids.data(), d_out, m * sizeof(uint32_t), hipMemcpyDeviceToHost);
  for (uint32_t i = 0; i < m; ++i) out_ids[i] = ids[i];
  hipFree(d_out);
  hipFree(d_cnt);
  return cnt;
}

// SpatialIndex::get_nearest_neighbours, location_hash_2d.rs:151-238.  Not on the
// step path: exact k-NN over a host copy,
// ties by ascending id.
size_t cs_query_knn(cs_engine* e, size_t k, double x, double y, uint64_t* out_ids) {
  hipSetDevice(e->device);
  cs_engine::HostState h;
  if (e->download(&h) != 0) return 0;
  std::vector<std::pair<double, uint64_t>> d;
  for (uint32_t i = 0; i < e->n_slots; ++i) {
    if (h.cell[i] == CS_INVALID_CELL) continue;
    double px, py;
    e->to_global(h.cell[i], h.off[i].x, h.off[i].y, &px, &py);
    d.push_back({std::sqrt((px - x) * (px - x) + (py - y) * (py - y)), h.id[i]});
  }
  std::sort(d.begin(), d.end());
  size_t n = std::min(k, d.size());
  for (size_t i = 0; i < n; ++i) out_ids[i] = d[i].second;
  return n;
}

void cs_profile_enable(cs_engine* e, uint32_t kernel_mask) { e->profiling = kernel_mask; }
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
  if (!e->tile || dir > 3 || capacity_records == 0 || capacity_records > 0x7FFFFFFFull) {
    e->error = "halo buffers need a tile engine, a direction 0..3 and a capacity";
    return 3;
  }
  e->halo[dir].send = static_cast<HaloRecord*>(send_dev);
  e->halo[dir].recv = static_cast<HaloRecord*>(recv_dev);
  e->halo[dir].cap = (uint32_t)capacity_records;
  // room for everything the four neighbours may deliver in one step
  return e->reserve((uint64_t)e->n_slots + 1024);
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

}  // extern "C"
