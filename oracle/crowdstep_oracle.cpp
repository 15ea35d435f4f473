// crowdstep_oracle.cpp — TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
//
// A CPU restatement (f64 by default) of the `Simulation::step` hot path of
// open-rmf/rmf_crowdsim, exported through the same C ABI as the HIP engine
// (include/crowdstep.h) so tests can drive both with one scenario.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
// library built from this file; the product (rmf_crowdsim_amd/) never does.
//
// Parity status: the reference is Rust and cannot be built or run in this
// pipeline (no rustc/cargo), so this file follows the cited
// lines by hand.  It is PINNED by the reference's own known answers
// (tests/test_oracle_reference_kats.py restates all eight reference tests):
// time_to_collision (zanlungo.rs:225-236), LocationHash2D radius / k-NN /
// update / remove (location_hash_2d.rs:311-397), one-step integration
// (lib.rs:423-453) and the 40-step source->sink stream
// (tests/event_listeners_test.rs:65-111).  The Zanlungo FORCE
// (compute_agent_force / right_of_way_vel / slerp) is pinned by no reference
// test: "parity unpinned" at that boundary, checked only against the
// hand-derived KAT-Z1..Z3 of SURVEY.md §8c.
//
// Shape: deliberately keeps the reference's data-structure choices (hash-map
// agent store with 88-byte Agent records, grid of per-cell id sets plus
// id->cell and id->position hash maps, a temporary vector per scanned cell, an
// Agent copy per neighbour, virtual planner calls behind a mutex, an update
// buffer plus commit), single-threaded like the reference (lib.rs:195,259), so
// that timing it is a fair "reference CPU path" column (cpu_baseline.kind =
// "port").  Two deliberate differences, both needed for a deterministic oracle
// (SURVEY.md §8a row a2):
//   * agents are visited in ascending id and ids inside a cell are kept in
//     ascending order (the reference uses RandomState HashMap/HashSet order);
//   * neighbour queries see start-of-step positions for every agent (Jacobi);
//     CS_ORACLE_GAUSS_SEIDEL=1 in the environment switches to the reference's
//     in-loop index update (lib.rs:299) in ascending-id order.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (oracle/Makefile).
// -DORACLE_REAL=float builds the f32 variant used to separate rounding growth
// from kernel bugs.

#include "../include/crowdstep.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <array>
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#ifndef ORACLE_REAL
#define ORACLE_REAL double
#endif
typedef ORACLE_REAL Real;

namespace {

struct V2 {
  Real x, y;
};
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, Real s) { return {a.x * s, a.y * s}; }
inline V2 operator-(V2 a) { return {-a.x, -a.y}; }
// nalgebra 0.31 Vector2<f64>: dot = x0*y0 + x1*y1, norm = sqrt(norm_squared),
// normalize = v / norm (call sites zanlungo.rs:50-52,112,159).
inline Real dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
inline Real norm2(V2 a) { return a.x * a.x + a.y * a.y; }
inline Real norm(V2 a) { return std::sqrt(norm2(a)); }
inline V2 normalized(V2 a) {
  Real n = norm(a);
  return {a.x / n, a.y / n};
}

const Real kInf = std::numeric_limits<Real>::infinity();

// Rust `f64 as usize`: saturating, NaN -> 0 (location_hash_2d.rs:56-57).
inline uint64_t sat_usize(Real v) {
  if (!(v > Real(0))) return 0;  // negative, -0, NaN
  if (v >= Real(18446744073709551615.0)) return UINT64_MAX;
  return (uint64_t)v;
}

// pub struct Agent, lib.rs:46-65 — 88 bytes like the reference's.
struct Agent {
  uint64_t agent_id;
  V2 position;
  Real orientation;
  V2 velocity;
  V2 preferred_vel;
  Real angular_vel;
  uint64_t next_waypoint;
  Real eyesight_range;
};

// ---------------------------------------------------------------------------
// LocationHash2D, spatial_index/location_hash_2d.rs:14-268
// ---------------------------------------------------------------------------
struct LocationHash2D {
  std::vector<std::set<uint64_t>> data;  // ascending ids inside a cell (canonical)
  std::unordered_map<uint64_t, uint64_t> id_to_index;
  std::unordered_map<uint64_t, V2> id_to_exact_location;
  Real width, height, resolution;
  V2 offset;

  // :33-51 — (width/cell as usize) * (height/cell as usize) cells
  LocationHash2D(Real w, Real h, Real cell, V2 off)
      : width(w), height(h), resolution(cell), offset(off) {
    uint64_t n = sat_usize(w / cell) * sat_usize(h / cell);
    data.resize(n);
  }

  uint64_t stride() const { return sat_usize(width / resolution); }

  // :54-66 — saturating casts, row stride width/res on BOTH axes, Err iff idx >= len
  bool location_to_index(V2 p, uint64_t* out) const {
    uint64_t xi = sat_usize((p - offset).x / resolution);
    uint64_t yi = sat_usize((p - offset).y / resolution);
    // usize arithmetic wraps in release builds; saturated inputs only occur for
    // absurd coordinates and land beyond len either way.
    unsigned __int128 idx = (unsigned __int128)xi * stride() + yi;
    if (idx >= data.size()) return false;
    *out = (uint64_t)idx;
    return true;
  }

  // :68-72
  void signed_xy(V2 p, int64_t* xi, int64_t* yi) const {
    Real fx = std::floor((p - offset).x / resolution);
    Real fy = std::floor((p - offset).y / resolution);
    // Rust `as i64` saturates; NaN -> 0
    auto sat = [](Real f) -> int64_t {
      if (f != f) return 0;
      if (f >= Real(9223372036854775807.0)) return INT64_MAX;
      if (f <= Real(-9223372036854775808.0)) return INT64_MIN;
      return (int64_t)f;
    };
    *xi = sat(fx);
    *yi = sat(fy);
  }

  // :74-85 — negative coordinates rejected, NO upper check on y, flat >= len rejected
  bool signed_to_data_idx(int64_t xi, int64_t yi, uint64_t* out) const {
    if (xi < 0 || yi < 0) return false;
    unsigned __int128 idx = (unsigned __int128)(uint64_t)xi * stride() + (uint64_t)yi;
    if (idx >= data.size()) return false;
    *out = (uint64_t)idx;
    return true;
  }

  // :87-101 — a fresh vector per scanned cell, one hash lookup per candidate
  bool neighbours_in_cell(int64_t xi, int64_t yi,
                          std::vector<std::pair<V2, uint64_t>>* out) const {
    uint64_t idx;
    if (!signed_to_data_idx(xi, yi, &idx)) return false;
    std::vector<std::pair<V2, uint64_t>> cell;
    for (uint64_t id : data[idx]) cell.push_back({id_to_exact_location.at(id), id});
    *out = std::move(cell);
    return true;
  }

  // :103-122
  void bounds(Real radius, V2 p, int64_t* l, int64_t* r, int64_t* b, int64_t* t) const {
    int64_t dummy;
    signed_xy({p.x + radius, p.y}, r, &dummy);
    signed_xy({p.x - radius, p.y}, l, &dummy);
    signed_xy({p.x, p.y + radius}, &dummy, t);
    signed_xy({p.x, p.y - radius}, &dummy, b);
  }

  // :126-149
  bool add_or_update(uint64_t id, V2 p) {
    uint64_t idx;
    if (!location_to_index(p, &idx)) return false;
    auto it = id_to_index.find(id);
    if (it != id_to_index.end()) {
      if (it->second != idx) {
        data[it->second].erase(id);
        data[idx].insert(id);
        it->second = idx;
      }
    } else {
      data[idx].insert(id);
      id_to_index[id] = idx;
    }
    id_to_exact_location[id] = p;
    return true;
  }

  // :240-258 — inclusive rectangle, x-major / y-minor, strict `<` on sqrt(dx²+dy²)
  std::vector<uint64_t> neighbours_in_radius(Real radius, V2 p) const {
    std::vector<uint64_t> found;
    int64_t l, r, b, t;
    bounds(radius, p, &l, &r, &b, &t);
    for (int64_t xi = l; xi <= r; ++xi) {
      for (int64_t yi = b; yi <= t; ++yi) {
        std::vector<std::pair<V2, uint64_t>> cell;
        if (!neighbours_in_cell(xi, yi, &cell)) continue;
        for (auto& c : cell)
          if (norm(c.first - p) < radius) found.push_back(c.second);
      }
    }
    return found;
  }

  // :151-238 — ring search exactly as written: half-open edge loops (the
  // (x+s, y+s) corner is never visited, (x-s, y-s) twice), stops once >= n
  // candidates were collected, stable sort by distance, first n.
  std::vector<uint64_t> nearest_neighbours(size_t n, V2 p) const {
    int64_t cx, cy;
    signed_xy(p, &cx, &cy);
    std::vector<std::pair<V2, uint64_t>> ring;
    bool all_out = false;
    int64_t s = 0;
    while (ring.size() < n && !all_out) {
      size_t out_of_bounds = 0, scanned = 0;
      auto visit = [&](int64_t xi, int64_t yi) {
        std::vector<std::pair<V2, uint64_t>> cell;
        if (neighbours_in_cell(xi, yi, &cell))
          ring.insert(ring.end(), cell.begin(), cell.end());
        else
          ++out_of_bounds;
        ++scanned;
      };
      if (s == 0) {
        visit(cx, cy);
      } else {
        for (int64_t i = cx - s; i < cx + s; ++i) visit(i, cy + s);  // top
        for (int64_t i = cx - s; i < cx + s; ++i) visit(i, cy - s);  // bottom
        for (int64_t i = cy - s; i < cy + s; ++i) visit(cx - s, i);  // left
        for (int64_t i = cy - s; i < cy + s; ++i) visit(cx + s, i);  // right
      }
      if (out_of_bounds == scanned) all_out = true;
      ++s;
    }
    std::stable_sort(ring.begin(), ring.end(),
                     [&](const std::pair<V2, uint64_t>& a, const std::pair<V2, uint64_t>& b) {
                       return norm(a.first - p) < norm(b.first - p);
                     });
    std::vector<uint64_t> ids;
    for (size_t i = 0; i < std::min(n, ring.size()); ++i) ids.push_back(ring[i].second);
    return ids;
  }

  // :260-267
  void remove_agent(uint64_t id) {
    auto it = id_to_index.find(id);
    if (it == id_to_index.end()) return;
    data[it->second].erase(id);
    id_to_exact_location.erase(id);
    id_to_index.erase(it);
  }
};

// ---------------------------------------------------------------------------
// Local planners: trait objects behind a mutex, as in lib.rs:79,288-291
// ---------------------------------------------------------------------------
struct LocalPlanner {
  virtual ~LocalPlanner() {}
  virtual V2 get_desired_velocity(const Agent& agent, const std::vector<Agent>& nearby,
                                  V2 recommended, bool* tti_zero) const = 0;
};

// no_local_plan.rs:9-17
struct NoLocalPlan : LocalPlanner {
  V2 get_desired_velocity(const Agent&, const std::vector<Agent>&, V2 recommended,
                          bool*) const override {
    return recommended;
  }
};

// any other `impl LocalPlanner` (local_planner.rs:7-18): host code behind the C ABI's batch callback, called
// here the way the reference calls a planner: once per agent, with its neighbours' old states (lib.rs:276-291)
struct CallbackLocalPlanner : LocalPlanner {
  cs_lp_batch_fn fn = nullptr;
  void* user = nullptr;
  mutable bool failed = false;  // the callback returned non-zero in this step: the step reports it (cs_step)
  static cs_lp_agent record(const Agent& a, bool with_preferred) {
    cs_lp_agent r;
    r.agent_id = a.agent_id;
    r.x = (double)a.position.x;
    r.y = (double)a.position.y;
    r.vx = (double)a.velocity.x;
    r.vy = (double)a.velocity.y;
    r.preferred_vx = with_preferred ? (double)a.preferred_vel.x : 0.0;
    r.preferred_vy = with_preferred ? (double)a.preferred_vel.y : 0.0;
    r.eyesight_range = (double)a.eyesight_range;
    r.next_waypoint = a.next_waypoint;
    return r;
  }
  V2 get_desired_velocity(const Agent& agent, const std::vector<Agent>& nearby, V2 recommended,
                          bool*) const override {
    const cs_lp_agent me = record(agent, true);
    std::vector<cs_lp_agent> nb;
    for (const Agent& a : nearby) nb.push_back(record(a, true));  // (a neighbour's preferred_vel is (0,0): lib.rs:140)
    const uint64_t begin[2] = {0, (uint64_t)nb.size()};
    if (nb.empty()) nb.emplace_back();
    const double rec[2] = {(double)recommended.x, (double)recommended.y};
    double out[2] = {0.0, 0.0};
    if (fn(user, 1, &me, rec, begin, nb.data(), out) != 0) failed = true;
    return V2{(Real)out[0], (Real)out[1]};
  }
};

// zanlungo.rs:9-218
struct Zanlungo : LocalPlanner {
  Real agent_scale, obstacle_scale, reaction_time, force_distance, agent_mass, agent_radius;
  std::unordered_map<uint64_t, Real> agent_priorities;  // :17,46 — created empty, no setter

  // Conditioning probe, NOT the reference (off by default; only oracle_fast_steps_guarded sets it): a pair that is not
  // inside the collision distance and whose |rel_vel|^2 is below 1e-30 counts as "never collides".  The reference
  // computes a collision 1e13 s or more ahead for such a pair (whose force term is exp(-huge) = 0) unless a and b^2
  // underflow, and then reads it as "colliding now" (spurious_collision below): in f64 that strikes once per 1e6
  // agent-steps, in a plain f32 transcription within tens of steps of any scene with real forces, so an f32 leg of a
  // long three-way comparison exists only with this guard (the engine has the same one: DESIGN.md section 2).
  bool guard_underflow = false;

  // :49-74
  Real time_to_collision(V2 rel_vel, V2 rel_pos) const {
    Real a = norm2(rel_vel);
    Real b = Real(2) * dot(rel_vel, rel_pos);
    Real c = norm2(rel_pos) - agent_radius * agent_radius;
    if (guard_underflow && c > Real(0) && a < Real(1e-30)) return kInf;
    Real disc = b * b - Real(4) * a * c;
    if (disc < Real(0)) return kInf;
    Real root = std::sqrt(disc);
    Real t0 = (-b - root) / (Real(2) * a);
    Real t1 = (-b + root) / (Real(2) * a);
    if ((t0 < Real(0) && t1 > Real(0)) || (t1 < Real(0) && t0 > Real(0))) return Real(0);
    if (t0 < t1 && t0 > Real(0)) return t0;
    if (t1 > Real(0)) return t1;
    return kInf;
  }

  // A diagnosis the reference does not make (test infrastructure: the parity tests name the agents they leave
  // out).  time_to_collision returns 0, "colliding now", for a pair that is NOT inside the collision distance when
  // |rel_vel|^2 underflows (f64: |rel_vel| ~ 1e-162) while b does not: root > |b| numerically, t0 = -inf, t1 = +inf.
  // The agent then gets 0/0 = NaN from its neighbours without right of way.  Such a pair is
  // told apart from a real overlap by c > 0.
  bool spurious_collision(V2 rel_vel, V2 rel_pos, Real t) const {
    return t == Real(0) && norm2(rel_pos) - agent_radius * agent_radius > Real(0) && norm2(rel_vel) < Real(1e-30);
  }
  mutable std::vector<uint64_t>* spurious_victims = nullptr;  // ids whose t_i came out 0 through such a pair
  mutable uint64_t* degenerate_flips = nullptr;               // (oracle_degenerate_flips)

  // :76-91 — strict `<` from +inf, all neighbours regardless of id
  Real compute_tti(const Agent& me, const std::vector<Agent>& nearby) const {
    Real t_i = kInf;
    for (const Agent& n : nearby) {
      Real t = time_to_collision(n.velocity - me.velocity, n.position - me.position);
      if (spurious_victims && spurious_collision(n.velocity - me.velocity, n.position - me.position, t))
        spurious_victims->push_back(me.agent_id);
      if (t < t_i) t_i = t;
    }
    return t_i;
  }

  Real priority_of(uint64_t id) const {
    auto it = agent_priorities.find(id);
    return it == agent_priorities.end() ? (Real)id : it->second;
  }

  // :173-198 — returns (signed sqrt of the clamped priority gap, my velocity,
  // other velocity), the lower-priority side blended toward its preferred velocity
  void right_of_way_vel(uint64_t my_id, V2 my_vel, V2 my_pref, V2 other_vel, V2 other_pref,
                        Real other_priority, Real* w, V2* out_my, V2* out_other) const {
    Real row = priority_of(my_id) - other_priority;
    if (row < Real(-1)) row = Real(-1);  // f64::clamp(-1, 1); NaN stays NaN
    if (row > Real(1)) row = Real(1);
    if (row < Real(0)) {
      Real r2 = std::sqrt(-row);
      *w = -r2;
      *out_my = my_vel;
      *out_other = other_vel + (other_pref - other_vel) * r2;
    } else if (row > Real(0)) {
      Real r2 = std::sqrt(row);
      *w = r2;
      *out_my = my_vel + (my_pref - my_vel) * r2;
      *out_other = other_vel;
    } else {
      *w = Real(0);
      *out_my = my_vel;
      *out_other = other_vel;
    }
  }

  // :23-28
  static V2 slerp(Real t, V2 p0, V2 p1, Real sin_theta) {
    Real theta = std::asin(sin_theta);
    Real t0 = std::sin((Real(1) - t) * theta) / sin_theta;
    Real t1 = std::sin(t * theta) / sin_theta;
    return p0 * t0 + p1 * t1;
  }

  // :93-170
  V2 compute_agent_force(const Agent& me, const Agent& other, Real t_i) const {
    Real w;
    V2 my_vel, other_vel;
    right_of_way_vel(me.agent_id, me.velocity, me.preferred_vel, other.velocity,
                     other.preferred_vel, priority_of(other.agent_id), &w, &my_vel, &other_vel);
    Real weight = Real(1) - w;
    V2 fut = me.position + my_vel * t_i;
    V2 other_fut = other.position + other_vel * t_i;
    V2 d = fut - other_fut;
    Real dist = norm(d);
    if (weight > Real(1)) {  // the other agent has right of way
      Real pref_speed = norm(other.preferred_vel);
      bool interpolate = true;
      V2 perp = {Real(0), Real(0)};
      if (pref_speed < Real(0.0001)) {
        // other wants to stand still: steer orthogonally to the current displacement
        V2 q = me.position - other.position;
        perp = {-q.y, q.x};
        // (certification of parity scenes: a neighbour straight ahead or behind makes this dot product zero in exact
        // arithmetic, its computed sign is rounding noise, and a 32-bit reading of the same line may flip the other
        // way: DESIGN.md section 5.  Counted while the term matters: finite t_i, the neighbour has right of way.)
        if (degenerate_flips && std::isfinite((double)t_i) &&
            std::fabs((double)dot(perp, me.velocity)) <= 1e-9 * (double)norm(perp) * (double)norm(me.velocity))
          ++*degenerate_flips;
        if (dot(perp, me.velocity) < Real(0)) perp = -perp;
      } else {
        // other is going somewhere: steer orthogonally to its preferred direction
        V2 pd = other.preferred_vel;
        if (dot(pd, d) > Real(0)) {
          perp = {-pd.y, pd.x};
          if (dot(perp, d) < Real(0)) perp = -perp;
        } else {
          interpolate = false;
        }
      }
      if (interpolate) {
        Real s = perp.x * d.y - perp.y * d.x;  // un-normalised determinant
        if (s < Real(0)) s = -s;
        if (s > Real(1)) s = Real(1);
        d = slerp(weight - Real(1), d, perp, s);
      }
    }
    // :155 — compares dist with the identical expression: never true
    if (dist > norm(fut - other_fut)) return {Real(0), Real(0)};
    V2 dn = normalized(d);
    Real surface = dist - agent_radius * Real(2);
    Real magnitude = weight * agent_scale * norm(my_vel - other_vel) / t_i;
    if (magnitude >= Real(1e15)) magnitude = Real(1e15);
    return dn * (magnitude * std::exp(-surface / force_distance));
  }

  // :201-217
  V2 get_desired_velocity(const Agent& agent, const std::vector<Agent>& nearby, V2 recommended,
                          bool* tti_zero) const override {
    Real t_i = compute_tti(agent, nearby);
    if (tti_zero) *tti_zero = (t_i == Real(0));
    V2 force = {Real(0), Real(0)};
    if (t_i != kInf)
      for (const Agent& n : nearby) force = force + compute_agent_force(agent, n, t_i);
    return recommended + force * (Real(1) / agent_mass);
  }
};

// ---------------------------------------------------------------------------
// High-level planners as data (include/crowdstep.h CS_HLP_*)
// ---------------------------------------------------------------------------
struct HighLevelPlanner {
  cs_hlp_desc d;
  // CS_HLP_ROUTE: RMFPlanner's bookkeeping, rmf/mod.rs:83-94 (the visibility graph and A*
  // are the host callback d.route_plan)
  std::map<uint64_t, std::pair<size_t, size_t>> agent_cache;  // id -> (route, next waypoint)
  std::vector<std::vector<V2>> route_list;
  std::map<std::array<long long, 4>, size_t> route_plans_by_location;
  static long long spatial_hash(double v, double res) {  // SpatialHash::new, rmf/mod.rs:70-77
    const double r = std::round(v / res);
    if (r != r) return 0;
    if (r >= 9.2e18) return INT64_MAX;
    if (r <= -9.2e18) return INT64_MIN;
    return (long long)r;
  }
  // highlevel_planners.rs:9 — Option<Vec2f>
  bool get_desired_velocity(const Agent& a, V2* out) {
    switch (d.kind) {
      case CS_HLP_ROUTE: {  // rmf/mod.rs:197-215
        auto it = agent_cache.find(a.agent_id);
        if (it == agent_cache.end()) return false;
        const std::vector<V2>& route = route_list[it->second.first];
        size_t wp = it->second.second;
        V2 dv = {a.position.x - route[wp].x, a.position.y - route[wp].y};
        if (std::sqrt(dv.x * dv.x + dv.y * dv.y) < (Real)d.route_arrive && route.size() > wp + 1) {
          wp += 1;
          it->second.second = wp;
        }
        V2 t = {route[wp].x - a.position.x, route[wp].y - a.position.y};
        const Real n = std::sqrt(t.x * t.x + t.y * t.y);  // normalize(): 0/0 = NaN on the waypoint
        *out = {t.x / n * (Real)d.route_speed, t.y / n * (Real)d.route_speed};
        return true;
      }
      case CS_HLP_CONSTANT:  // lib.rs:403-410
        *out = {(Real)d.vx, (Real)d.vy};
        return true;
      case CS_HLP_ID_PARITY:  // rmf_crowdsim_viz/src/main.rs:20-30
        if (a.agent_id % 2 == 0)
          *out = {(Real)-d.vx, (Real)-d.vy};
        else
          *out = {(Real)d.vx, (Real)d.vy};
        return true;
      case CS_HLP_CALLBACK: {
        double pos[2] = {(double)a.position.x, (double)a.position.y};
        double vel[2] = {(double)a.velocity.x, (double)a.velocity.y};
        double o[2] = {0, 0};
        uint8_t some = 0;
        uint64_t id = a.agent_id;
        if (d.velocity) d.velocity(d.user, 1, &id, pos, vel, 0.0, o, &some);
        *out = {(Real)o[0], (Real)o[1]};
        return some != 0;
      }
      default:
        return false;
    }
  }
  void set_target(const Agent& a, V2 point, V2 tol) {
    if (d.kind == CS_HLP_CALLBACK && d.set_target)
      d.set_target(d.user, a.agent_id, (double)a.position.x, (double)a.position.y,
                   (double)point.x, (double)point.y, (double)tol.x, (double)tol.y);
    if (d.kind == CS_HLP_ROUTE && d.route_plan) {  // rmf/mod.rs:217-236
      const double sx = (double)a.position.x, sy = (double)a.position.y;
      const std::array<long long, 4> key = {spatial_hash(sx, d.route_scale), spatial_hash(sy, d.route_scale),
                                            spatial_hash((double)point.x, d.route_scale),
                                            spatial_hash((double)point.y, d.route_scale)};
      auto it = route_plans_by_location.find(key);
      if (it != route_plans_by_location.end()) {
        agent_cache[a.agent_id] = {it->second, 0};
        return;
      }
      std::vector<double> buf(2 * CS_ROUTE_MAX_WAYPOINTS);
      size_t n = d.route_plan(d.user, sx, sy, (double)point.x, (double)point.y, buf.data(),
                              CS_ROUTE_MAX_WAYPOINTS);
      n = std::min<size_t>(n, CS_ROUTE_MAX_WAYPOINTS);
      if (n == 0) return;  // "Failed to find contiguous path between source and target"
      std::vector<V2> route(n);
      for (size_t k = 0; k < n; ++k) route[k] = {(Real)buf[2 * k], (Real)buf[2 * k + 1]};
      route_plans_by_location[key] = route_list.size();
      agent_cache[a.agent_id] = {route_list.size(), 0};
      route_list.push_back(route);
    }
  }
  void remove_agent_id(uint64_t id) {
    if (d.kind == CS_HLP_CALLBACK && d.remove_agent) d.remove_agent(d.user, id);
    if (d.kind == CS_HLP_ROUTE) agent_cache.erase(id);  // rmf/mod.rs:239-241
  }
};

// ---------------------------------------------------------------------------
// Crowd generators, source_sink.rs:30-33,75-82,96-100
// ---------------------------------------------------------------------------
inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// Seeded stand-in for PoissonCrowd (its thread_rng cannot be seeded): Knuth's
// product method on uniforms from splitmix64(seed, step, draw).
inline uint64_t poisson_seeded(uint64_t seed, uint64_t step, double mean) {
  if (!(mean > 0.0)) return 0;
  double limit = std::exp(-mean), prod = 1.0;
  uint64_t k = 0;
  for (uint64_t draw = 0; draw < 1000000; ++draw) {
    uint64_t r = splitmix64(seed ^ splitmix64(step * 0x100000001B3ull + draw));
    double u = (double)((r >> 11) + 1) * (1.0 / 9007199254740993.0);  // (0,1)
    prod *= u;
    if (prod <= limit) break;
    ++k;
  }
  return k;
}

struct SourceSink {  // source_sink.rs:36-60
  V2 source;
  Real radius_sink;
  uint32_t generator_kind;
  double rate;
  uint64_t seed;
  cs_generator_fn generator;
  void* generator_user;
  uint32_t hlp, lp;
  std::vector<V2> waypoints;
  bool loop_forever;
  Real agent_eyesight_range;
  uint64_t calls = 0;

  uint64_t number_to_spawn(double dt) {
    uint64_t step = calls++;
    switch (generator_kind) {
      case CS_GEN_MONOTONIC: {  // source_sink.rs:96-100: (dt*rate).round() as usize
        double v = std::round(dt * rate);
        return v > 0 ? (uint64_t)v : 0;
      }
      case CS_GEN_POISSON_SEEDED:
        return poisson_seeded(seed, step, dt * rate);
      case CS_GEN_CALLBACK:
        return generator ? (uint64_t)generator(generator_user, dt) : 0;
    }
    return 0;
  }
};

struct StateUpdate {  // lib.rs:94-99
  V2 new_vel, new_pos;
  bool updated;
  uint64_t next_waypoint;
};

}  // namespace

// ---------------------------------------------------------------------------
// Simulation<LocationHash2D>, lib.rs:69-383
// ---------------------------------------------------------------------------
struct cs_engine {
  std::unordered_map<uint64_t, Agent> agents;  // lib.rs:71
  std::vector<uint64_t> order;                 // ascending ids (canonical visiting order)
  std::unordered_map<uint32_t, std::shared_ptr<SourceSink>> source_sinks;  // Registry, registry.rs
  uint32_t next_sink_handle = 0;
  LocationHash2D index;
  std::unordered_map<uint64_t, uint32_t> agent_hlp, agent_lp;  // lib.rs:77,79
  uint64_t last_alloc_agent_id = 0;                            // lib.rs:83
  std::unordered_map<uint64_t, StateUpdate> update_buffer;     // lib.rs:86
  std::unordered_map<uint64_t, uint32_t> correspondence;       // lib.rs:90
  std::vector<std::shared_ptr<LocalPlanner>> lps;
  std::vector<std::shared_ptr<HighLevelPlanner>> hlps;
  std::vector<cs_snapshot_record> snapshot;
  uint64_t snapshot_step = 0, steps_done = 0;
  bool snapshot_valid = false;
  std::mutex planner_lock;  // stands in for the per-planner Mutex (lib.rs:264-268,288-291)
  std::vector<cs_event> events;
  bool record_events = true;
  std::string error;
  bool gauss_seidel = false;
  // ordered pairs (i, j) whose membership in i's radius query depends on
  // whether j's index entry is still the old position or already the new one, i.e. where the
  // reference's in-loop index update (lib.rs:299) could make it differ from the Jacobi result
  bool count_shell = false;
  std::vector<uint64_t> spurious_victims;  // (oracle_spurious_victims)
  uint64_t degenerate_flips = 0;           // (oracle_degenerate_flips)
  uint64_t last_shell_crossings = 0;

  explicit cs_engine(const cs_grid_desc& g)
      : index((Real)g.width, (Real)g.height, (Real)g.cell_size,
              V2{(Real)g.offset_x, (Real)g.offset_y}) {
    const char* gs = std::getenv("CS_ORACLE_GAUSS_SEIDEL");
    gauss_seidel = gs && gs[0] == '1';
  }

  // lib.rs:119-156
  bool add_agents(const std::vector<V2>& pts, uint32_t hlp, uint32_t lp, Real eyesight,
                  uint32_t owner, std::vector<uint64_t>* out) {
    for (const V2& p : pts) {
      uint64_t id = last_alloc_agent_id++;
      agent_hlp[id] = hlp;
      agent_lp[id] = lp;
      Agent a;
      a.agent_id = id;
      a.position = p;
      a.orientation = 0;
      a.velocity = {0, 0};
      a.preferred_vel = {0, 0};
      a.angular_vel = 0;
      a.next_waypoint = 0;
      a.eyesight_range = eyesight;
      agents[id] = a;
      order.push_back(id);  // ids only grow, so `order` stays sorted
      if (!index.add_or_update(id, p)) {
        error = "Index out of bounds";
        return false;  // earlier agents stay added, as in the reference
      }
      out->push_back(id);
      cs_event e;
      e.kind = CS_EVENT_SPAWNED;
      e.source_sink = owner;
      e.id = id;
      e.x = (double)p.x;
      e.y = (double)p.y;
      if (record_events) events.push_back(e);
    }
    return true;
  }

  // lib.rs:176-192
  void remove_agent(uint64_t id) {
    hlps[agent_hlp.at(id)]->remove_agent_id(id);
    agents.erase(id);
    order.erase(std::lower_bound(order.begin(), order.end(), id));
    update_buffer.erase(id);
    correspondence.erase(id);
    index.remove_agent(id);
    cs_event e;
    e.kind = CS_EVENT_DESTROYED;
    e.source_sink = UINT32_MAX;
    e.id = id;
    e.x = e.y = 0;
    if (record_events) events.push_back(e);
  }

  // lib.rs:195-383
  int step(double dt_seconds, cs_step_report* rep) {
    cs_step_report r;
    std::memset(&r, 0, sizeof r);
    const Real dt = (Real)dt_seconds;

    // ---- Phase A: spawn (lib.rs:199-254) ---------------------------------
    // pass 1: every sink's occupancy test against the index as the previous
    // step left it; ascending handle stands in for HashMap order.
    std::vector<uint32_t> handles;
    for (auto& kv : source_sinks) handles.push_back(kv.first);
    std::sort(handles.begin(), handles.end());
    std::vector<std::pair<uint32_t, std::vector<V2>>> to_add;
    for (uint32_t h : handles) {
      SourceSink& s = *source_sinks[h];
      uint64_t n = s.number_to_spawn(dt_seconds);
      std::vector<V2> pts;
      if (n > 0) {  // the loop over n is commented out in the reference (lib.rs:207)
        if (index.neighbours_in_radius(Real(0.4), s.source).empty()) pts.push_back(s.source);
      }
      to_add.push_back({h, pts});
    }
    // pass 2: all adds
    std::vector<std::pair<uint32_t, std::pair<bool, std::vector<uint64_t>>>> added;
    for (auto& ta : to_add) {
      SourceSink& s = *source_sinks[ta.first];
      std::vector<uint64_t> ids;
      bool ok = add_agents(ta.second, s.hlp, s.lp, s.agent_eyesight_range, ta.first, &ids);
      added.push_back({ta.first, {ok, ids}});
    }
    // pass 3: correspondence + set_target
    for (auto& ad : added) {
      if (!ad.second.first) {
        error = "Failed to add agents from source";
        return 1;
      }
      SourceSink& s = *source_sinks[ad.first];
      for (uint64_t id : ad.second.second) {
        correspondence[id] = ad.first;
        hlps[agent_hlp[id]]->set_target(agents[id], s.waypoints[0],
                                        V2{s.radius_sink, s.radius_sink});
        ++r.n_spawned;
      }
    }

    // ---- Phase B: per-agent update (lib.rs:259-347) -----------------------
    std::vector<uint64_t> to_be_removed;
    std::vector<std::pair<uint64_t, V2>> deferred_index_updates;
    for (uint64_t agent_id : order) {
      Agent agent = agents.at(agent_id);  // clone of the OLD state (lib.rs:261)
      V2 vel = {0, 0};
      {
        std::lock_guard<std::mutex> g(planner_lock);
        V2 v;
        if (hlps[agent_hlp[agent_id]]->get_desired_velocity(agent, &v)) {
          vel = v;
          agent.preferred_vel = v;  // only the local clone carries it (lib.rs:271)
        }
      }
      {
        std::vector<uint64_t> ids =
            index.neighbours_in_radius(agent.eyesight_range, agent.position);
        std::vector<Agent> neighbours;  // 88-byte copies of OLD states (lib.rs:281-286)
        for (uint64_t nid : ids)
          if (nid != agent_id) neighbours.push_back(agents.at(nid));
        std::lock_guard<std::mutex> g(planner_lock);
        bool tz = false;
        vel = lps[agent_lp[agent_id]]->get_desired_velocity(agent, neighbours, vel, &tz);
        if (tz) ++r.n_tti_zero;
      }
      V2 new_pos = agent.position + vel * dt;  // lib.rs:295-297
      if (!(std::isfinite((double)new_pos.x) && std::isfinite((double)new_pos.y) &&
            std::isfinite((double)vel.x) && std::isfinite((double)vel.y)))
        ++r.n_nonfinite;
      if ((new_pos - index.offset).x < 0 || (new_pos - index.offset).y < 0) ++r.n_clamped;

      if (gauss_seidel) {
        if (!index.add_or_update(agent_id, new_pos)) {  // lib.rs:299-302
          error = "Index out of bounds";
          return 1;
        }
      } else {
        uint64_t idx;
        if (!index.location_to_index(new_pos, &idx)) {
          error = "Index out of bounds";
          return 1;  // nothing committed; index untouched (DESIGN.md "errors")
        }
        deferred_index_updates.push_back({agent_id, new_pos});
      }

      // waypoint / sink test on the OLD position (lib.rs:304-336)
      uint64_t next_waypoint = agent.next_waypoint;
      auto corr = correspondence.find(agent_id);
      // An agent whose source-sink was removed (lib.rs:164-168) makes the reference panic here:
      // `registry[source_sink_id]` on a missing key (lib.rs:307).  The engine (and this oracle)
      // let such an agent walk on without waypoint tests instead.
      if (corr != correspondence.end() && source_sinks.count(corr->second)) {
        SourceSink& s = *source_sinks.at(corr->second);
        if (agent.next_waypoint >= s.waypoints.size()) {
          // "Rogue agent": the reference pushes it for removal and then indexes
          // out of range (panic).  Unreachable through this ABI; treated as removal.
          to_be_removed.push_back(agent_id);
        } else if (norm(agent.position - s.waypoints[agent.next_waypoint]) < s.radius_sink) {
          ++r.n_waypoint_hits;
          if (agent.next_waypoint == s.waypoints.size() - 1) {
            if (s.loop_forever)
              next_waypoint = 0;
            else
              to_be_removed.push_back(agent_id);
          } else {
            next_waypoint += 1;
            hlps[agent_hlp[agent_id]]->set_target(agents.at(agent_id), s.waypoints[next_waypoint],
                                                  V2{s.radius_sink, s.radius_sink});
          }
        }
      }
      update_buffer[agent_id] = StateUpdate{vel, new_pos, true, next_waypoint};
    }
    if (count_shell) {
      std::unordered_map<uint64_t, V2> moved;
      Real dmax = 0;
      for (auto& u : deferred_index_updates) {
        moved[u.first] = u.second;
        const V2 d = u.second - agents.at(u.first).position;
        const Real n = std::sqrt(d.x * d.x + d.y * d.y);
        if (n == n && n > dmax) dmax = n;
      }
      last_shell_crossings = 0;
      for (uint64_t i : order) {
        const Agent& me = agents.at(i);
        const Real r = me.eyesight_range;
        for (uint64_t j : index.neighbours_in_radius(r + dmax, me.position)) {
          if (j == i) continue;
          const V2 po = agents.at(j).position - me.position;
          auto it = moved.find(j);
          if (it == moved.end()) continue;
          const V2 pn = it->second - me.position;
          const bool in_old = std::sqrt(po.x * po.x + po.y * po.y) < r;
          const bool in_new = std::sqrt(pn.x * pn.x + pn.y * pn.y) < r;
          if (in_old != in_new) ++last_shell_crossings;
        }
      }
    }
    for (auto& u : deferred_index_updates) index.add_or_update(u.first, u.second);

    // ---- Phase C: commit (lib.rs:350-359) ---------------------------------
    for (auto& kv : update_buffer) {
      if (!kv.second.updated) continue;
      Agent& a = agents.at(kv.first);
      a.velocity = kv.second.new_vel;
      a.position = kv.second.new_pos;
      a.next_waypoint = kv.second.next_waypoint;
      kv.second.updated = false;
    }

    // ---- Phase D: removal (lib.rs:378-380) --------------------------------
    for (uint64_t id : to_be_removed) {
      remove_agent(id);
      ++r.n_destroyed;
    }
    r.n_agents = agents.size();
    if (rep) *rep = r;
    return 0;
  }
};

// ---------------------------------------------------------------------------
// C ABI (include/crowdstep.h)
// ---------------------------------------------------------------------------
extern "C" {

uint32_t cs_abi_version(void) { return CS_ABI_VERSION; }

cs_engine* cs_create(const cs_grid_desc* grid, const cs_device_cfg*) {
  if (!grid) return nullptr;
  return new cs_engine(*grid);
}
void cs_destroy(cs_engine* e) { delete e; }
const char* cs_last_error(const cs_engine* e) { return e ? e->error.c_str() : "null engine"; }
const char* cs_backend_name(const cs_engine*) {
  return sizeof(Real) == 8 ? "oracle:f64" : "oracle:f32";
}

uint32_t cs_register_zanlungo(cs_engine* e, const cs_zanlungo_params* p) {
  auto z = std::make_shared<Zanlungo>();
  z->agent_scale = (Real)p->agent_scale;
  z->obstacle_scale = (Real)p->obstacle_scale;
  z->reaction_time = (Real)p->reaction_time;
  z->force_distance = (Real)p->force_distance;
  z->agent_mass = (Real)p->agent_mass;
  z->agent_radius = (Real)p->agent_radius;
  e->lps.push_back(z);
  return (uint32_t)e->lps.size() - 1;
}
uint32_t cs_register_no_local_plan(cs_engine* e) {
  e->lps.push_back(std::make_shared<NoLocalPlan>());
  return (uint32_t)e->lps.size() - 1;
}
uint32_t cs_register_lp_callback(cs_engine* e, cs_lp_batch_fn fn, void* user) {
  if (!fn) {
    e->error = "cs_register_lp_callback: the planner's function is null";
    return UINT32_MAX;
  }
  auto p = std::make_shared<CallbackLocalPlanner>();
  p->fn = fn;
  p->user = user;
  e->lps.push_back(p);
  return (uint32_t)e->lps.size() - 1;
}
uint32_t cs_register_hlp(cs_engine* e, const cs_hlp_desc* d) {
  auto h = std::make_shared<HighLevelPlanner>();
  h->d = *d;
  e->hlps.push_back(h);
  return (uint32_t)e->hlps.size() - 1;
}

int cs_add_agents(cs_engine* e, const double* xy, size_t n, uint32_t hlp, uint32_t lp,
                  double eyesight, uint64_t* out_ids) {
  if (hlp >= e->hlps.size() || lp >= e->lps.size()) {
    e->error = "unknown planner handle";
    return 2;
  }
  std::vector<V2> pts(n);
  for (size_t i = 0; i < n; ++i) pts[i] = {(Real)xy[2 * i], (Real)xy[2 * i + 1]};
  std::vector<uint64_t> ids;
  bool ok = e->add_agents(pts, hlp, lp, (Real)eyesight, UINT32_MAX, &ids);
  if (out_ids)
    for (size_t i = 0; i < ids.size(); ++i) out_ids[i] = ids[i];
  return ok ? 0 : 1;
}

int cs_remove_agent(cs_engine* e, uint64_t id) {
  if (!e->agents.count(id)) {
    e->error = "unknown agent id";
    return 2;
  }
  e->remove_agent(id);
  return 0;
}

uint32_t cs_add_source_sink(cs_engine* e, const cs_source_sink_desc* d) {
  auto s = std::make_shared<SourceSink>();
  s->source = {(Real)d->source_x, (Real)d->source_y};
  s->radius_sink = (Real)d->radius_sink;
  s->generator_kind = d->generator_kind;
  s->rate = d->rate;
  s->seed = d->seed;
  s->generator = d->generator;
  s->generator_user = d->generator_user;
  s->hlp = d->hlp;
  s->lp = d->lp;
  for (size_t i = 0; i < d->n_waypoints; ++i)
    s->waypoints.push_back({(Real)d->waypoints_xy[2 * i], (Real)d->waypoints_xy[2 * i + 1]});
  s->loop_forever = d->loop_forever != 0;
  s->agent_eyesight_range = (Real)d->agent_eyesight_range;
  uint32_t h = e->next_sink_handle++;  // registry.rs:16-21
  e->source_sinks[h] = s;
  return h;
}
void cs_remove_source_sink(cs_engine* e, uint32_t handle) { e->source_sinks.erase(handle); }
size_t cs_source_sink_slots(cs_engine* e) { return e->next_sink_handle; }
uint64_t cs_device_bytes(cs_engine*) { return 0; }
uint64_t cs_kernel_stat(cs_engine*, uint32_t) { return 0; }  // (the HIP engine's own diagnostics)
// batch forms: the single queries in a loop (location_hash_2d.rs:240-258, :151-238)
size_t cs_query_radius(cs_engine* e, double radius, double x, double y, uint64_t* out_ids, size_t cap);
size_t cs_query_knn(cs_engine* e, size_t k, double x, double y, uint64_t* out_ids);
int cs_query_radius_batch(cs_engine* e, size_t n, const double* xy, const double* radius, size_t cap_per_query,
                          uint64_t* out_ids, uint64_t* out_counts, float*, uint32_t*) {
  for (size_t i = 0; i < n; ++i) {
    const size_t c = cs_query_radius(e, radius[i], xy[2 * i], xy[2 * i + 1], out_ids + i * cap_per_query, cap_per_query);
    if (out_counts) out_counts[i] = c;
  }
  return 0;
}
int cs_query_knn_batch(cs_engine* e, size_t n, const double* xy, size_t k, uint64_t* out_ids, uint64_t* out_counts,
                       float*) {
  for (size_t i = 0; i < n; ++i) {
    std::vector<uint64_t> tmp(k + e->agents.size() + 1);
    const size_t c = std::min(k, cs_query_knn(e, k, xy[2 * i], xy[2 * i + 1], tmp.data()));
    for (size_t j = 0; j < c; ++j) out_ids[i * k + j] = tmp[j];
    if (out_counts) out_counts[i] = c;
  }
  return 0;
}

int cs_step(cs_engine* e, double dt_seconds, cs_step_report* report) {
  int rc = e->step(dt_seconds, report);
  for (auto& lp : e->lps)
    if (auto* cb = dynamic_cast<CallbackLocalPlanner*>(lp.get()))
      if (cb->failed) {
        cb->failed = false;
        e->error = "a host LocalPlanner failed (its callback returned non-zero)";
        rc = 9;
      }
  if (rc == 0) e->steps_done += 1;
  return rc;
}
// the streaming view has nothing to overlap on the CPU: a plain copy of `agents` (lib.rs:71)
int cs_snapshot_request(cs_engine* e) {
  e->snapshot.clear();
  for (uint64_t id : e->order) {
    const Agent& a = e->agents.at(id);
    cs_snapshot_record r;
    r.x = (double)a.position.x;
    r.y = (double)a.position.y;
    r.vx = (float)a.velocity.x;
    r.vy = (float)a.velocity.y;
    r.id = (uint32_t)id;
    r.next_waypoint = (uint32_t)a.next_waypoint;
    e->snapshot.push_back(r);
  }
  e->snapshot_step = e->steps_done;
  e->snapshot_valid = true;
  return 0;
}
int cs_snapshot_acquire(cs_engine* e, int, const cs_snapshot_record** out, size_t* n, uint64_t* step_index) {
  if (!e->snapshot_valid) return 1;
  *out = e->snapshot.data();
  *n = e->snapshot.size();
  if (step_index) *step_index = e->snapshot_step;
  return 0;
}
int cs_synchronize(cs_engine*) { return 0; }

size_t cs_agent_count(cs_engine* e) { return e->agents.size(); }

size_t cs_read_agents(cs_engine* e, cs_agent_view* out, size_t cap) {
  size_t n = 0;
  for (uint64_t id : e->order) {
    if (n >= cap) break;
    const Agent& a = e->agents.at(id);
    out[n].id = id;
    out[n].x = (double)a.position.x;
    out[n].y = (double)a.position.y;
    out[n].vx = (double)a.velocity.x;
    out[n].vy = (double)a.velocity.y;
    out[n].next_waypoint = a.next_waypoint;
    out[n].eyesight_range = (double)a.eyesight_range;
    ++n;
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

size_t cs_query_radius(cs_engine* e, double radius, double x, double y, uint64_t* out_ids,
                       size_t cap) {
  std::vector<uint64_t> ids = e->index.neighbours_in_radius((Real)radius, V2{(Real)x, (Real)y});
  for (size_t i = 0; i < std::min(cap, ids.size()); ++i) out_ids[i] = ids[i];
  return ids.size();
}

size_t cs_query_knn(cs_engine* e, size_t k, double x, double y, uint64_t* out_ids) {
  std::vector<uint64_t> ids = e->index.nearest_neighbours(k, V2{(Real)x, (Real)y});
  for (size_t i = 0; i < ids.size(); ++i) out_ids[i] = ids[i];
  return ids.size();
}

void cs_profile_enable(cs_engine*, uint32_t) {}
void cs_profile_stride(cs_engine*, uint32_t) {}
int cs_profile_read(cs_engine*, uint32_t, double* total_ms, uint64_t* launches) {
  if (total_ms) *total_ms = 0;
  if (launches) *launches = 0;
  return 0;
}
void cs_profile_reset(cs_engine*) {}
int cs_halo_set_buffers(cs_engine* e, uint32_t, void*, void*, uint64_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_halo_pack(cs_engine* e, uint32_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_halo_unpack(cs_engine* e, uint32_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_halo_pack_all(cs_engine* e) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_halo_unpack_all(cs_engine* e) {
  e->error = "oracle has no tiles";
  return 3;
}
size_t cs_spawn_probe(cs_engine* e, double, uint8_t*, size_t) {
  e->error = "oracle has no tiles";
  return SIZE_MAX;
}
int cs_spawn_commit(cs_engine* e, const uint8_t*, size_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_spawn_probe_dev(cs_engine* e, double, int*, size_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_spawn_commit_dev(cs_engine* e, const int*, size_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_tile_histogram(cs_engine* e, uint64_t*, uint64_t*) {
  e->error = "oracle has no tiles";
  return 3;
}
size_t cs_tile_export(cs_engine*, void*, size_t) { return SIZE_MAX; }
int cs_tile_retile(cs_engine* e, uint32_t, uint32_t, uint32_t, uint32_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_tile_import(cs_engine* e, const void*, size_t) {
  e->error = "oracle has no tiles";
  return 3;
}
size_t cs_route_misses(cs_engine*, cs_route_miss*, size_t) { return 0; }
int cs_route_resolve(cs_engine* e, const cs_route_miss*, size_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_rccl_unique_id(uint8_t*) { return 8; }
int cs_rccl_comm_init(cs_engine* e, int32_t, int32_t, const uint8_t*) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_rccl_comm_adopt(cs_engine* e, void*) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_halo_set_peers(cs_engine* e, const int32_t*) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_halo_exchange_rccl(cs_engine* e, int32_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_allreduce_max_i32_rccl(cs_engine* e, int*, size_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_allgather_bytes_rccl(cs_engine* e, const void*, void*, size_t) {
  e->error = "oracle has no tiles";
  return 3;
}
int cs_tile_step_rccl(cs_engine* e, double, cs_step_report*) {
  e->error = "oracle has no tiles";
  return 3;
}

// ---- cs_mesh_*: the reference is ONE process, so the oracle's "mesh" is one reference
// simulation whatever the tiling asked for: the checker of a tile mesh is the untiled crowd.
struct cs_mesh {
  cs_engine* e = nullptr;
  std::string error;
};
static thread_local std::string g_mesh_error;
cs_mesh* cs_mesh_create(const cs_grid_desc* grid, const cs_mesh_desc* d) {
  if (!grid || !d || d->tiles_x == 0 || d->tiles_y == 0 || d->halo_cells == 0) {
    g_mesh_error = "cs_mesh_create: a grid, tiles_x, tiles_y >= 1 and halo_cells >= 1 are required";
    return nullptr;
  }
  cs_engine* e = cs_create(grid, nullptr);
  if (!e) {
    g_mesh_error = cs_last_error(nullptr);
    return nullptr;
  }
  cs_mesh* m = new cs_mesh();
  m->e = e;
  return m;
}
void cs_mesh_destroy(cs_mesh* m) {
  if (!m) return;
  cs_destroy(m->e);
  delete m;
}
const char* cs_mesh_last_error(const cs_mesh* m) { return m ? cs_last_error(m->e) : g_mesh_error.c_str(); }
size_t cs_mesh_local_tiles(const cs_mesh*) { return 1; }
cs_engine* cs_mesh_tile(cs_mesh* m, size_t k) { return k == 0 ? m->e : nullptr; }
int cs_mesh_tile_rect(const cs_mesh* m, size_t k, uint32_t* r) {
  if (k) return 3;
  r[0] = 0; r[1] = (uint32_t)sat_usize(m->e->index.height / m->e->index.resolution); r[2] = 0; r[3] = (uint32_t)m->e->index.stride();
  return 0;
}
uint32_t cs_mesh_register_zanlungo(cs_mesh* m, const cs_zanlungo_params* p) { return cs_register_zanlungo(m->e, p); }
uint32_t cs_mesh_register_no_local_plan(cs_mesh* m) { return cs_register_no_local_plan(m->e); }
uint32_t cs_mesh_register_hlp(cs_mesh* m, const cs_hlp_desc* d) { return cs_register_hlp(m->e, d); }
uint32_t cs_mesh_register_lp_callback(cs_mesh* m, cs_lp_batch_fn fn, void* user) { return cs_register_lp_callback(m->e, fn, user); }
int cs_mesh_add_agents(cs_mesh* m, const double* xy, size_t n, uint32_t hlp, uint32_t lp, double eyesight, uint64_t* out) {
  return cs_add_agents(m->e, xy, n, hlp, lp, eyesight, out);
}
uint32_t cs_mesh_add_source_sink(cs_mesh* m, const cs_source_sink_desc* d) { return cs_add_source_sink(m->e, d); }
void cs_mesh_remove_source_sink(cs_mesh* m, uint32_t h) { cs_remove_source_sink(m->e, h); }
int cs_mesh_remove_agent(cs_mesh* m, uint64_t id) { return cs_remove_agent(m->e, id); }
void cs_mesh_event_recording(cs_mesh* m, int on) { cs_event_recording(m->e, on); }
size_t cs_mesh_drain_events(cs_mesh* m, cs_event* out, size_t cap) { return cs_drain_events(m->e, out, cap); }
int cs_mesh_step(cs_mesh* m, double dt, cs_step_report* r) { return cs_step(m->e, dt, r); }
int cs_mesh_synchronize(cs_mesh* m) { return cs_synchronize(m->e); }
size_t cs_mesh_agent_count(cs_mesh* m) { return cs_agent_count(m->e); }
size_t cs_mesh_read_agents(cs_mesh* m, cs_agent_view* out, size_t cap) { return cs_read_agents(m->e, out, cap); }
int cs_mesh_tile_counts(cs_mesh* m, uint64_t* out) {
  out[0] = cs_agent_count(m->e);
  return 0;
}
uint64_t cs_mesh_exchange_bytes(const cs_mesh*) { return 0; }  // (one process: nothing travels)
int cs_mesh_recut(cs_mesh*) { return 0; }  // (nothing to cut)
int cs_mesh_query_radius_batch(cs_mesh* m, size_t n, const double* xy, const double* radius, size_t cap, uint64_t* out_ids,
                               uint64_t* out_counts) {
  return cs_query_radius_batch(m->e, n, xy, radius, cap, out_ids, out_counts, nullptr, nullptr);
}
int cs_mesh_query_knn_batch(cs_mesh* m, size_t n, const double* xy, size_t k, uint64_t* out_ids, uint64_t* out_counts) {
  return cs_query_knn_batch(m->e, n, xy, k, out_ids, out_counts, nullptr);
}

// Oracle-only probes used by tests/test_oracle_reference_kats.py to pin the
// private pieces the reference's own unit tests reach (zanlungo.rs:225-236).
double oracle_time_to_collision(double agent_radius, double rvx, double rvy, double rpx,
                                double rpy) {
  Zanlungo z;
  z.agent_scale = 1;
  z.obstacle_scale = 1;
  z.reaction_time = 0;
  z.force_distance = 1;
  z.agent_mass = 1;
  z.agent_radius = (Real)agent_radius;
  return (double)z.time_to_collision(V2{(Real)rvx, (Real)rvy}, V2{(Real)rpx, (Real)rpy});
}

// Oracle-only probes for tests/test_zanlungo_restatement.py: the planner on explicit Agent
// records, INCLUDING a neighbour's preferred_vel (always (0,0) inside Simulation::step,
// lib.rs:140,261,271, so the moving-neighbour branch zanlungo.rs:126-139 is reachable only here).
// rec = {id, px, py, vx, vy, pref_x, pref_y}; params = Zanlungo::new's six arguments in order.
static Zanlungo probe_planner(const double* params) {
  Zanlungo z;
  z.agent_scale = (Real)params[0];
  z.obstacle_scale = (Real)params[1];
  z.reaction_time = (Real)params[2];
  z.force_distance = (Real)params[3];
  z.agent_mass = (Real)params[4];
  z.agent_radius = (Real)params[5];
  return z;
}
static Agent probe_agent(const double* rec) {
  Agent a = Agent{};
  a.agent_id = (uint64_t)rec[0];
  a.position = {(Real)rec[1], (Real)rec[2]};
  a.velocity = {(Real)rec[3], (Real)rec[4]};
  a.preferred_vel = {(Real)rec[5], (Real)rec[6]};
  return a;
}
void oracle_zanlungo_pair_force(const double* params, const double* me, const double* other, double t_i,
                                double* out_xy) {
  const Zanlungo z = probe_planner(params);
  const V2 f = z.compute_agent_force(probe_agent(me), probe_agent(other), (Real)t_i);
  out_xy[0] = (double)f.x;
  out_xy[1] = (double)f.y;
}
double oracle_zanlungo_desired_velocity(const double* params, const double* me, const double* others,
                                        uint64_t n_others, double rec_x, double rec_y, double* out_xy) {
  const Zanlungo z = probe_planner(params);
  std::vector<Agent> nearby;
  for (uint64_t k = 0; k < n_others; ++k) nearby.push_back(probe_agent(others + 7 * k));
  const Agent a = probe_agent(me);
  const V2 v = z.get_desired_velocity(a, nearby, V2{(Real)rec_x, (Real)rec_y}, nullptr);
  out_xy[0] = (double)v.x;
  out_xy[1] = (double)v.y;
  return (double)z.compute_tti(a, nearby);
}

// SpatialIndex::add_or_update / remove_agent on the engine's index without
// creating an Agent (location_hash_2d.rs:371-397 exercises the index alone).
int oracle_index_add_or_update(cs_engine* e, uint64_t id, double x, double y) {
  return e->index.add_or_update(id, V2{(Real)x, (Real)y}) ? 0 : 1;
}
void oracle_index_remove(cs_engine* e, uint64_t id) { e->index.remove_agent(id); }
// certification of a parity scene (row a2): count, for the steps that follow, the neighbour pairs
// that sit on the eyesight shell during the step; read the last step's count back
void oracle_count_shell_crossings(cs_engine* e, int on) { e->count_shell = on != 0; }
// ids of the agents whose t_i came out 0 through an underflowed pair so far (Zanlungo::spurious_collision), with
// repeats; recorded from the call on.  Returns the number of records (may exceed cap).
size_t oracle_spurious_victims(cs_engine* e, uint64_t* out, size_t cap) {
  for (auto& lp : e->lps)
    if (auto* z = dynamic_cast<Zanlungo*>(lp.get())) z->spurious_victims = &e->spurious_victims;
  for (size_t k = 0; k < e->spurious_victims.size() && k < cap; ++k) out[k] = e->spurious_victims[k];
  return e->spurious_victims.size();
}
// force terms so far whose sideways direction hung on the sign of a dot product that is zero to rounding (a
// neighbour straight ahead or behind: Zanlungo::compute_agent_force); counted from the first call on
uint64_t oracle_degenerate_flips(cs_engine* e) {
  for (auto& lp : e->lps)
    if (auto* z = dynamic_cast<Zanlungo*>(lp.get())) z->degenerate_flips = &e->degenerate_flips;
  return e->degenerate_flips;
}
uint64_t oracle_shell_crossings(cs_engine* e) { return e->last_shell_crossings; }

// ---------------------------------------------------------------------------
// "What a good CPU does" (SURVEY.md section 8d): the same canonical step with the same Zanlungo
// arithmetic (the methods above), but on cell-sorted arrays instead of the reference's hash
// maps, and with the agent loop spread over threads (OpenMP).  NOT the reference's shape: bench.py
// reports it beside the reference-shaped port, clearly labelled.  Crowd of agents with fixed
// preferred velocities (the bench scene), no source-sinks.  Positions are updated in place;
// returns the seconds spent in the step loop, or a negative value if an agent left the grid.
// ---------------------------------------------------------------------------
}  // extern "C"
// flags of oracle_fast_steps_ex (test infrastructure; 0 = the reference's arithmetic as restated above)
//   1  Zanlungo::guard_underflow
//   2  "cell-relative": positions are KEPT in f64 whatever Real is, and every agent's update is computed in Real on
//      positions taken relative to the origin of the agent's own cell (rounded to Real once).  The model only ever uses
//      differences of positions, so in exact arithmetic nothing changes; in an f32 build the rounding of a coordinate is
//      then that of a number below a few cells (1e-7 m) instead of that of a global coordinate (3e-5 m at 500 m).  This
//      is the precision class of the HIP engine's layout in an independent implementation: the
//      f32 leg of the long three-way comparisons.
static double fast_steps_impl(uint64_t n, double* xy, double* vel_xy, const double* pref_xy, double agent_scale,
                              double force_distance, double agent_mass, double agent_radius, double eyesight,
                              double width, double height, double cell_size, double off_x, double off_y,
                              double dt_seconds, uint32_t steps, int threads, uint8_t* spurious_out, int flags) {
  // spurious_out (may be null): set to 1 for every agent whose t_i came out 0 through an underflowed pair in some
  // step (Zanlungo::spurious_collision)
  const bool relative = (flags & 2) != 0;
  Zanlungo lp;
  lp.guard_underflow = (flags & 1) != 0;
  lp.agent_scale = (Real)agent_scale;
  lp.obstacle_scale = Real(1);
  lp.reaction_time = Real(0);
  lp.force_distance = (Real)force_distance;
  lp.agent_mass = (Real)agent_mass;
  lp.agent_radius = (Real)agent_radius;
  const uint64_t nx = (uint64_t)(width / cell_size), n_rows = (uint64_t)(height / cell_size);
  const uint64_t ncells = nx * n_rows;
  std::vector<Agent> cur(n), nxt(n);
  std::vector<double> pos(relative ? 2 * n : 0), pos_nxt(relative ? 2 * n : 0);  // (flag 2: the positions proper)
  for (uint64_t i = 0; i < n; ++i) {
    Agent& a = cur[i];
    a = Agent{};
    a.agent_id = i;
    a.position = {(Real)xy[2 * i], (Real)xy[2 * i + 1]};
    a.velocity = {(Real)vel_xy[2 * i], (Real)vel_xy[2 * i + 1]};
    a.eyesight_range = (Real)eyesight;
    if (relative) pos[2 * i] = xy[2 * i], pos[2 * i + 1] = xy[2 * i + 1];
  }
  auto px = [&](uint64_t i) { return relative ? pos[2 * i] : (double)cur[i].position.x; };
  auto py = [&](uint64_t i) { return relative ? pos[2 * i + 1] : (double)cur[i].position.y; };
  std::vector<uint32_t> cell_of(n), start(ncells + 1), order(n);
  bool failed = false;
  const auto t0 = std::chrono::steady_clock::now();
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  for (uint32_t s = 0; s < steps && !failed; ++s) {
    // cells in the reference's layout (x_idx * nx + y_idx); members in ascending id
    std::fill(start.begin(), start.end(), 0u);
    for (uint64_t i = 0; i < n; ++i) {
      const double fx = (px(i) - off_x) / cell_size, fy = (py(i) - off_y) / cell_size;
      const uint64_t cx = fx > 0 ? (uint64_t)fx : 0, cy = fy > 0 ? (uint64_t)fy : 0;
      const uint64_t flat = cx * nx + cy;
      if (flat >= ncells) {
        failed = true;
        break;
      }
      cell_of[i] = (uint32_t)flat;
      start[flat + 1] += 1;
    }
    if (failed) break;
    for (uint64_t c = 0; c < ncells; ++c) start[c + 1] += start[c];
    {
      std::vector<uint32_t> fill(start.begin(), start.end() - 1);
      for (uint64_t i = 0; i < n; ++i) order[fill[cell_of[i]]++] = (uint32_t)i;  // ids ascending inside a cell
    }
    const Real dt = (Real)dt_seconds;
#pragma omp parallel
    {
      std::vector<const Agent*> nearby;
      std::vector<Agent> local;  // (flag 2: the neighbours' records with positions relative to this agent's cell)
#pragma omp for schedule(dynamic, 256)
      for (long long k = 0; k < (long long)n; ++k) {
        const uint32_t i = order[k];  // walk the agents cell by cell: neighbours stay in cache
        Agent me = cur[i];
        me.preferred_vel = {(Real)pref_xy[2 * i], (Real)pref_xy[2 * i + 1]};
        const Real r = me.eyesight_range;
        const double mx = px(i), my = py(i);
        // the origin everything is measured from under flag 2: this agent's cell
        const double ox = relative ? off_x + std::floor((mx - off_x) / cell_size) * cell_size : 0.0;
        const double oy = relative ? off_y + std::floor((my - off_y) / cell_size) * cell_size : 0.0;
        if (relative) me.position = {(Real)(mx - ox), (Real)(my - oy)};
        const long long lx = (long long)std::floor((mx - (double)r - off_x) / cell_size);
        const long long hx = (long long)std::floor((mx + (double)r - off_x) / cell_size);
        const long long ly = (long long)std::floor((my - (double)r - off_y) / cell_size);
        const long long hy = (long long)std::floor((my + (double)r - off_y) / cell_size);
        nearby.clear();
        local.clear();
        for (long long x = lx; x <= hx; ++x)
          for (long long y = ly; y <= hy; ++y) {
            if (x < 0 || y < 0) continue;
            const unsigned long long flat = (unsigned long long)x * nx + (unsigned long long)y;
            if (flat >= ncells) continue;
            for (uint32_t q = start[flat]; q < start[flat + 1]; ++q) {
              const uint32_t j = order[q];
              const Agent& o = cur[j];
              if (o.agent_id == me.agent_id) continue;
              if (!relative) {
                if (norm(o.position - me.position) < r) nearby.push_back(&o);
              } else {
                Agent rel = o;
                rel.position = {(Real)(pos[2 * j] - ox), (Real)(pos[2 * j + 1] - oy)};
                if (norm(rel.position - me.position) < r) local.push_back(rel);
              }
            }
          }
        if (relative)
          for (const Agent& o : local) nearby.push_back(&o);
        Real t_i = kInf;
        for (const Agent* o : nearby) {
          const Real t = lp.time_to_collision(o->velocity - me.velocity, o->position - me.position);
          if (spurious_out && lp.spurious_collision(o->velocity - me.velocity, o->position - me.position, t))
            spurious_out[i] = 1;  // (this agent's own byte: no race)
          if (t < t_i) t_i = t;
        }
        V2 force = {Real(0), Real(0)};
        if (t_i != kInf)
          for (const Agent* o : nearby) force = force + lp.compute_agent_force(me, *o, t_i);
        const V2 vel = me.preferred_vel + force * (Real(1) / lp.agent_mass);
        Agent out = cur[i];
        out.velocity = vel;
        if (!relative) {
          out.position = me.position + vel * dt;
        } else {
          // the step itself in Real relative to the cell (as the engine's layout does), kept in f64
          const V2 moved = me.position + vel * dt;
          pos_nxt[2 * i] = ox + (double)moved.x;
          pos_nxt[2 * i + 1] = oy + (double)moved.y;
          out.position = {(Real)pos_nxt[2 * i], (Real)pos_nxt[2 * i + 1]};
        }
        nxt[i] = out;
      }
    }
    cur.swap(nxt);
    if (relative) pos.swap(pos_nxt);
  }
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (uint64_t i = 0; i < n; ++i) {
    xy[2 * i] = px(i);
    xy[2 * i + 1] = py(i);
    vel_xy[2 * i] = (double)cur[i].velocity.x;
    vel_xy[2 * i + 1] = (double)cur[i].velocity.y;
  }
  return failed ? -1.0 : el;
}
extern "C" {
double oracle_fast_steps(uint64_t n, double* xy, double* vel_xy, const double* pref_xy, double agent_scale,
                         double force_distance, double agent_mass, double agent_radius, double eyesight,
                         double width, double height, double cell_size, double off_x, double off_y,
                         double dt_seconds, uint32_t steps, int threads, uint8_t* spurious_out) {
  return fast_steps_impl(n, xy, vel_xy, pref_xy, agent_scale, force_distance, agent_mass, agent_radius, eyesight, width,
                         height, cell_size, off_x, off_y, dt_seconds, steps, threads, spurious_out, 0);
}
// The same with the flags above (the f32 build's leg of the long three-way comparisons: flags = 3).
double oracle_fast_steps_ex(uint64_t n, double* xy, double* vel_xy, const double* pref_xy, double agent_scale,
                            double force_distance, double agent_mass, double agent_radius, double eyesight,
                            double width, double height, double cell_size, double off_x, double off_y,
                            double dt_seconds, uint32_t steps, int threads, uint8_t* spurious_out, int flags) {
  return fast_steps_impl(n, xy, vel_xy, pref_xy, agent_scale, force_distance, agent_mass, agent_radius, eyesight, width,
                         height, cell_size, off_x, off_y, dt_seconds, steps, threads, spurious_out, flags);
}
}  // extern "C"
