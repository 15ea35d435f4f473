// crowdsim.hpp — C++ host-side mirror of the reference's trait surface over the C ABI.
//
// Same names, argument order and error behaviour as rmf_crowdsim (paths relative to
// rmf_crowdsim/src in the reference tree); Result<_, String> becomes std::runtime_error(message).
//   Simulation            lib.rs:69-383            EventListener     lib.rs:22-33
//   HighLevelPlanner      highlevel_planners/highlevel_planners.rs:8-16
//   LocalPlanner / Zanlungo / NoLocalPlan   local_planners/{local_planner,zanlungo,no_local_plan}.rs
//   LocationHash2D        spatial_index/location_hash_2d.rs:33-51
//   SourceSink / CrowdGenerator / MonotonicCrowd / PoissonCrowd   source_sink/source_sink.rs:30-101
// Header-only; link against libcrowdstep_hip.so (the HIP engine).  There is no CPU path.
#ifndef CROWDSIM_HPP
#define CROWDSIM_HPP

#include <chrono>
#include <map>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "crowdstep.h"

namespace rmf_crowdsim {

using AgentId = std::size_t;  // lib.rs:36
struct Vec2f {               // lib.rs:40-43 (nalgebra Vector2<f64>)
  double x = 0, y = 0;
};
using Point = Vec2f;

struct Agent {  // lib.rs:46-65
  AgentId agent_id = 0;
  Point position;
  double orientation = 0;
  Vec2f velocity;
  double angular_vel = 0;
  std::size_t next_waypoint = 0;
  double eyesight_range = 0;
};

struct EventListener {  // lib.rs:22-33
  virtual ~EventListener() = default;
  virtual void agent_spawned(Vec2f position, AgentId agent) = 0;
  virtual void agent_destroyed(AgentId agent) = 0;
  virtual void waypoint_reached(Vec2f, AgentId) {}
};

struct LocationHash2D;
// trait SpatialIndex, spatial_index.rs:4-14: what Simulation<T: SpatialIndex> (lib.rs:69) is generic over.  On this
// backend the index IS the neighbour kernel (the engine keeps the agents in a LocationHash2D's cell order and answers the
// trait's four calls itself), so an index has to describe itself as such a grid: device_form() returns the LocationHash2D
// it is equivalent to, or null, in which case Simulation's constructor throws.  No host-side slow path for a foreign
// index exists (a per-agent host query per step would put the hot path on the CPU).
struct SpatialIndex {
  virtual ~SpatialIndex() = default;
  virtual const LocationHash2D* device_form() const { return nullptr; }
};
struct LocationHash2D : SpatialIndex {  // location_hash_2d.rs:33
  double width, height, cell_size;
  Point offset;
  LocationHash2D(double w, double h, double cell, Point off) : width(w), height(h), cell_size(cell), offset(off) {}
  const LocationHash2D* device_form() const override { return this; }
};

// highlevel_planners.rs:8-16.  Override get_desired_velocity for a host planner (batched
// callback, slow path), or use the data planners below, which the device evaluates.
struct HighLevelPlanner {
  virtual ~HighLevelPlanner() = default;
  virtual bool get_desired_velocity(const Agent&, std::chrono::duration<double>, Vec2f*) { return false; }
  virtual void set_target(const Agent&, Point, Vec2f) {}
  virtual void remove_agent_id(AgentId) {}
  virtual cs_hlp_desc describe() {
    cs_hlp_desc d{};
    d.kind = CS_HLP_CALLBACK;
    d.user = this;
    d.velocity = [](void* u, size_t n, const uint64_t* ids, const double* pos, const double* vel, double t,
                    double* out, uint8_t* some) {
      auto* self = static_cast<HighLevelPlanner*>(u);
      for (size_t i = 0; i < n; ++i) {
        Agent a;
        a.agent_id = ids[i];
        a.position = {pos[2 * i], pos[2 * i + 1]};
        a.velocity = {vel[2 * i], vel[2 * i + 1]};
        Vec2f v;
        some[i] = self->get_desired_velocity(a, std::chrono::duration<double>(t), &v) ? 1 : 0;
        out[2 * i] = v.x;
        out[2 * i + 1] = v.y;
      }
    };
    d.set_target = [](void* u, uint64_t id, double px, double py, double tx, double ty, double ox, double oy) {
      Agent a;
      a.agent_id = id;
      a.position = {px, py};
      static_cast<HighLevelPlanner*>(u)->set_target(a, {tx, ty}, {ox, oy});
    };
    d.remove_agent = [](void* u, uint64_t id) { static_cast<HighLevelPlanner*>(u)->remove_agent_id(id); };
    return d;
  }
};
struct StubHighLevelPlan : HighLevelPlanner {  // lib.rs:391-420: Some(default_vel)
  Vec2f default_vel;
  explicit StubHighLevelPlan(Vec2f v) : default_vel(v) {}
  cs_hlp_desc describe() override {
    cs_hlp_desc d{};
    d.kind = CS_HLP_CONSTANT;
    d.vx = default_vel.x;
    d.vy = default_vel.y;
    return d;
  }
};
struct IdParityHighLevelPlan : StubHighLevelPlan {  // rmf_crowdsim_viz/src/main.rs:20-30
  using StubHighLevelPlan::StubHighLevelPlan;
  cs_hlp_desc describe() override {
    cs_hlp_desc d = StubHighLevelPlan::describe();
    d.kind = CS_HLP_ID_PARITY;
    return d;
  }
};
// The follower half of RMFPlanner (rmf/mod.rs:195-242): override plan_route (the A* of
// rmf/mod.rs:160-192 is the host's business); it is asked once per new (start, goal) hash pair,
// following the route runs on the device.  An empty result = no contiguous path.
struct RouteFollower : HighLevelPlanner {
  double scale = 1.0, arrive = 0.1, speed = 1.0;
  virtual std::vector<Point> plan_route(Point start, Point goal) = 0;
  cs_hlp_desc describe() override {
    cs_hlp_desc d{};
    d.kind = CS_HLP_ROUTE;
    d.user = this;
    d.route_scale = scale;
    d.route_arrive = arrive;
    d.route_speed = speed;
    d.route_plan = [](void* u, double sx, double sy, double gx, double gy, double* out, size_t cap) -> size_t {
      const std::vector<Point> r = static_cast<RouteFollower*>(u)->plan_route({sx, sy}, {gx, gy});
      const size_t n = r.size() < cap ? r.size() : cap;
      for (size_t k = 0; k < n; ++k) {
        out[2 * k] = r[k].x;
        out[2 * k + 1] = r[k].y;
      }
      return n;
    };
    return d;
  }
};

// local_planner.rs:7-18.  The shipped planners (below) run on the device; any other subclass is host code:
// override get_desired_velocity and the engine evaluates it every step through its batched callback
// (cs_register_lp_callback: the slow path), with the agent and its neighbours as they were at the start of the step.
struct LocalPlanner {
  virtual ~LocalPlanner() = default;
  virtual Vec2f get_desired_velocity(const Agent& agent, const std::vector<Agent>& nearby_agents,
                                     Vec2f recommended_velocity) const {
    (void)agent; (void)nearby_agents;
    return recommended_velocity;
  }
  static int batch_thunk(void* u, size_t n, const cs_lp_agent* agents, const double* rec, const uint64_t* nb_begin,
                         const cs_lp_agent* nb, double* out) {
          try {
          const auto view = [](const cs_lp_agent& r) {
            Agent a{};
            a.agent_id = r.agent_id;
            a.position = {r.x, r.y};
            a.velocity = {r.vx, r.vy};
            a.next_waypoint = r.next_waypoint;
            a.eyesight_range = r.eyesight_range;
            return a;
          };
          const LocalPlanner* self = static_cast<const LocalPlanner*>(u);
          for (size_t k = 0; k < n; ++k) {
            std::vector<Agent> nearby;
            for (uint64_t q = nb_begin[k]; q < nb_begin[k + 1]; ++q) nearby.push_back(view(nb[q]));
            const Vec2f v = self->get_desired_velocity(view(agents[k]), nearby, Vec2f{rec[2 * k], rec[2 * k + 1]});
            out[2 * k] = v.x;
            out[2 * k + 1] = v.y;
          }
          } catch (...) {  // (nothing may unwind through the C frame: the step fails instead)
            return 1;
          }
          return 0;
  }
  virtual uint32_t register_with(cs_engine* e) { return cs_register_lp_callback(e, &LocalPlanner::batch_thunk, this); }
  // (a tile mesh evaluates planners on every tile: each tile asks for the agents it owns)
  virtual uint32_t register_with(cs_mesh* m) { return cs_mesh_register_lp_callback(m, &LocalPlanner::batch_thunk, this); }
};
struct NoLocalPlan : LocalPlanner {  // no_local_plan.rs:7-18
  uint32_t register_with(cs_engine* e) override { return cs_register_no_local_plan(e); }
  uint32_t register_with(cs_mesh* m) override { return cs_mesh_register_no_local_plan(m); }
};
struct Zanlungo : LocalPlanner {  // zanlungo.rs:31-48
  cs_zanlungo_params p;
  Zanlungo(double agent_scale, double obstacle_scale, double reaction_time, double force_distance,
           double agent_mass, double agent_radius)
      : p{agent_scale, obstacle_scale, reaction_time, force_distance, agent_mass, agent_radius} {}
  uint32_t register_with(cs_engine* e) override { return cs_register_zanlungo(e, &p); }
  uint32_t register_with(cs_mesh* m) override { return cs_mesh_register_zanlungo(m, &p); }
};

struct CrowdGenerator {  // source_sink.rs:30-33
  virtual ~CrowdGenerator() = default;
  virtual std::size_t get_number_to_spawn(std::chrono::duration<double> time_elapsed) const = 0;
  virtual void fill(cs_source_sink_desc* d) const {
    d->generator_kind = CS_GEN_CALLBACK;
    d->generator_user = const_cast<CrowdGenerator*>(this);
    d->generator = [](void* u, double dt) {
      return static_cast<const CrowdGenerator*>(u)->get_number_to_spawn(std::chrono::duration<double>(dt));
    };
  }
};
struct MonotonicCrowd : CrowdGenerator {  // source_sink.rs:85-101
  double rate;
  explicit MonotonicCrowd(double r) : rate(r) {}
  std::size_t get_number_to_spawn(std::chrono::duration<double> t) const override {
    double v = t.count() * rate;
    v = v < 0 ? 0 : (double)(long long)(v + 0.5);
    return (std::size_t)v;
  }
  void fill(cs_source_sink_desc* d) const override {
    d->generator_kind = CS_GEN_MONOTONIC;
    d->rate = rate;
  }
};

struct PoissonCrowd : CrowdGenerator {  // source_sink.rs:63-82: unseeded like the reference's thread_rng; host callback
  double rate;
  explicit PoissonCrowd(double r) : rate(r) {}
  std::size_t get_number_to_spawn(std::chrono::duration<double> t) const override {
    const double rt = t.count() * rate;
    if (!(rt > 0.0)) return 0;  // (the reference panics here: Poisson::new(rt).unwrap(), source_sink.rs:79; no exception may
                                //  cross the C ABI this is called through)
    thread_local std::mt19937_64 rng{std::random_device{}()};
    return (std::size_t)std::poisson_distribution<unsigned long long>(rt)(rng);
  }
};

struct SourceSink {  // source_sink.rs:36-60
  Vec2f source;
  double radius_sink;
  std::shared_ptr<CrowdGenerator> crowd_generator;
  std::shared_ptr<HighLevelPlanner> high_level_planner;
  std::shared_ptr<LocalPlanner> local_planner;
  std::vector<Vec2f> waypoints;
  bool loop_forever;
  double agent_eyesight_range;
};

class Simulation {  // Simulation<LocationHash2D>, lib.rs:69-383
 public:
  std::unordered_map<AgentId, Agent> agents;  // lib.rs:71, refreshed after every mutating call

  // Simulation<T: SpatialIndex>::new (lib.rs:103) for any index that describes itself as a grid
  explicit Simulation(const SpatialIndex& any_index, int device = 0) : Simulation(as_grid(any_index), device) {}
  static const LocationHash2D& as_grid(const SpatialIndex& index) {
    const LocationHash2D* grid = index.device_form();
    if (!grid)
      throw std::runtime_error("Simulation<T: SpatialIndex>: on this backend the spatial index is the neighbour kernel itself; "
                               "this index does not describe itself as a uniform grid (device_form() -> LocationHash2D)");
    return *grid;
  }
  explicit Simulation(const LocationHash2D& index, int device = 0) {  // lib.rs:103
    cs_grid_desc g{index.width, index.height, index.cell_size, index.offset.x, index.offset.y};
    cs_device_cfg cfg{};
    cfg.device_ordinal = device;
    engine_ = cs_create(&g, &cfg);
    if (!engine_) throw std::runtime_error(std::string("cs_create failed: ") + cs_last_error(nullptr));
  }
  ~Simulation() { cs_destroy(engine_); }
  Simulation(const Simulation&) = delete;
  Simulation& operator=(const Simulation&) = delete;

  std::vector<AgentId> add_agents(const std::vector<Point>& spawn_positions,  // lib.rs:119-156
                                  std::shared_ptr<HighLevelPlanner> hlp, std::shared_ptr<LocalPlanner> lp,
                                  double agent_eyesight_range) {
    std::vector<double> xy;
    for (const Point& p : spawn_positions) {
      xy.push_back(p.x);
      xy.push_back(p.y);
    }
    std::vector<uint64_t> ids(spawn_positions.size());
    int rc = cs_add_agents(engine_, xy.data(), ids.size(), handle(hlp), handle(lp), agent_eyesight_range,
                           ids.data());
    after_mutation();
    if (rc != 0) throw std::runtime_error(cs_last_error(engine_));
    return std::vector<AgentId>(ids.begin(), ids.end());
  }
  std::size_t add_source_sink(std::shared_ptr<SourceSink> s) {  // lib.rs:159-161
    cs_source_sink_desc d{};
    d.source_x = s->source.x;
    d.source_y = s->source.y;
    d.radius_sink = s->radius_sink;
    s->crowd_generator->fill(&d);
    d.hlp = handle(s->high_level_planner);
    d.lp = handle(s->local_planner);
    std::vector<double> wps;
    for (const Vec2f& w : s->waypoints) {
      wps.push_back(w.x);
      wps.push_back(w.y);
    }
    d.waypoints_xy = wps.data();
    d.n_waypoints = s->waypoints.size();
    d.loop_forever = s->loop_forever ? 1 : 0;
    d.agent_eyesight_range = s->agent_eyesight_range;
    sinks_.push_back(s);
    const uint32_t sink = cs_add_source_sink(engine_, &d);
    if (sink == UINT32_MAX) throw std::runtime_error(cs_last_error(engine_));
    return sink;
  }
  void remove_source_sink(std::size_t id) { cs_remove_source_sink(engine_, (uint32_t)id); }  // lib.rs:164
  std::size_t add_event_listener(std::shared_ptr<EventListener> l) {                        // lib.rs:171
    listeners_[next_listener_] = std::move(l);
    return next_listener_++;
  }
  void remove_agents(AgentId agent) {  // lib.rs:176-192 (unknown id throws instead of panicking)
    int rc = cs_remove_agent(engine_, agent);
    after_mutation();
    if (rc != 0) throw std::runtime_error(cs_last_error(engine_));
  }
  void step(std::chrono::duration<double> dur) {  // lib.rs:195-383
    cs_step_report rep;
    int rc = cs_step(engine_, dur.count(), &rep);
    after_mutation();
    if (rc != 0) throw std::runtime_error(cs_last_error(engine_));
  }
  // The same step without refreshing `agents` and without waiting for the device (an embedder
  // that renders from snapshots; listeners still get their events at the next step() / refresh).
  void step_no_readback(std::chrono::duration<double> dur) {
    if (cs_step(engine_, dur.count(), nullptr) != 0) throw std::runtime_error(cs_last_error(engine_));
  }
  // Streaming `agents` view (rmf_crowdsim_viz/src/main.rs:112-128): request a frame behind the
  // queued steps, pick it up later; the records stay valid until the second request from now.
  void request_snapshot() {
    if (cs_snapshot_request(engine_) != 0) throw std::runtime_error(cs_last_error(engine_));
  }
  struct Frame {
    const cs_snapshot_record* agents = nullptr;
    std::size_t count = 0;
    uint64_t step_index = 0;
    bool ready = false;
  };
  Frame snapshot(bool wait = true) {
    Frame f;
    int rc = cs_snapshot_acquire(engine_, wait ? 1 : 0, &f.agents, &f.count, &f.step_index);
    if (rc == 3) throw std::runtime_error(cs_last_error(engine_));
    f.ready = rc == 0;
    return f;
  }

 private:
  template <class P>
  uint32_t handle(const std::shared_ptr<P>& p) {
    auto it = handles_.find(p.get());
    if (it != handles_.end()) return it->second;
    uint32_t h = register_planner(p.get());
    handles_[p.get()] = h;
    keep_.push_back(p);
    return h;
  }
  uint32_t register_planner(HighLevelPlanner* p) {
    cs_hlp_desc d = p->describe();
    return cs_register_hlp(engine_, &d);
  }
  uint32_t register_planner(LocalPlanner* p) { return p->register_with(engine_); }

  void after_mutation() {
    cs_event ev[256];
    for (;;) {
      std::size_t n = cs_drain_events(engine_, ev, 256);
      for (std::size_t i = 0; i < n; ++i)
        for (auto& kv : listeners_) {
          if (ev[i].kind == CS_EVENT_SPAWNED) kv.second->agent_spawned({ev[i].x, ev[i].y}, ev[i].id);
          if (ev[i].kind == CS_EVENT_DESTROYED) kv.second->agent_destroyed(ev[i].id);
        }
      if (n < 256) break;
    }
    std::vector<cs_agent_view> v(cs_agent_count(engine_));
    std::size_t got = cs_read_agents(engine_, v.data(), v.size());
    agents.clear();
    for (std::size_t i = 0; i < got; ++i) {
      Agent a;
      a.agent_id = v[i].id;
      a.position = {v[i].x, v[i].y};
      a.velocity = {v[i].vx, v[i].vy};
      a.next_waypoint = v[i].next_waypoint;
      a.eyesight_range = v[i].eyesight_range;
      agents[a.agent_id] = a;
    }
  }

  cs_engine* engine_ = nullptr;
  std::map<const void*, uint32_t> handles_;
  std::vector<std::shared_ptr<void>> keep_;
  std::vector<std::shared_ptr<SourceSink>> sinks_;
  std::map<std::size_t, std::shared_ptr<EventListener>> listeners_;
  std::size_t next_listener_ = 0;
};

// The same `Simulation`, cut into tiles_x x tiles_y spatial tiles: one tile engine per tile behind
// the C ABI's mesh handle (cs_mesh_*: layout, halo exchange, spawn flags, route misses, re-cuts, merged queries all
// in the library).  With `rccl_unique_id` (CS_RCCL_UNIQUE_ID_BYTES from cs_rccl_unique_id on one rank) the
// distributed form: one tile per rank and GPU; a `host_transport` (three functions over MPI, sockets, ...:
// cs_mesh_host_transport) can stand in for RCCL or beside it.  In the distributed forms every call is collective and
// `agents`, re-cuts and the queries cover the whole crowd on every rank.  Results equal Simulation's bit for bit while nobody touches the
// domain's edges.
class TiledSimulation {
 public:
  std::unordered_map<AgentId, Agent> agents;  // lib.rs:71, refreshed after every mutating call

  TiledSimulation(const LocationHash2D& index, uint32_t tiles_x, uint32_t tiles_y, uint32_t halo_cells, int device = 0,
                  const uint8_t* rccl_unique_id = nullptr, int rank = 0, int n_ranks = 1, double density_per_cell = 16.0,
                  const cs_mesh_host_transport* host_transport = nullptr) {
    cs_grid_desc g{index.width, index.height, index.cell_size, index.offset.x, index.offset.y};
    cs_mesh_desc d{};
    d.tiles_x = tiles_x;
    d.tiles_y = tiles_y;
    d.halo_cells = halo_cells;
    d.device_ordinal = device;
    d.rank = rank;
    d.n_ranks = n_ranks;
    d.density_per_cell = density_per_cell;
    d.rccl_unique_id = rccl_unique_id;
    d.host_transport = host_transport;
    mesh_ = cs_mesh_create(&g, &d);
    if (!mesh_) throw std::runtime_error(std::string("cs_mesh_create failed: ") + cs_mesh_last_error(nullptr));
  }
  ~TiledSimulation() { cs_mesh_destroy(mesh_); }
  TiledSimulation(const TiledSimulation&) = delete;
  TiledSimulation& operator=(const TiledSimulation&) = delete;

  std::vector<AgentId> add_agents(const std::vector<Point>& spawn_positions, std::shared_ptr<HighLevelPlanner> hlp,
                                  std::shared_ptr<LocalPlanner> lp, double agent_eyesight_range) {  // lib.rs:119-156
    std::vector<double> xy;
    for (const Point& p : spawn_positions) {
      xy.push_back(p.x);
      xy.push_back(p.y);
    }
    std::vector<uint64_t> ids(spawn_positions.size());
    const int rc = cs_mesh_add_agents(mesh_, xy.data(), ids.size(), handle(hlp), handle(lp), agent_eyesight_range, ids.data());
    refresh();
    if (rc != 0) throw std::runtime_error(cs_mesh_last_error(mesh_));
    return std::vector<AgentId>(ids.begin(), ids.end());
  }
  std::size_t add_source_sink(std::shared_ptr<SourceSink> s) {  // lib.rs:159-161
    cs_source_sink_desc d{};
    d.source_x = s->source.x;
    d.source_y = s->source.y;
    d.radius_sink = s->radius_sink;
    s->crowd_generator->fill(&d);
    d.hlp = handle(s->high_level_planner);
    d.lp = handle(s->local_planner);
    std::vector<double> wps;
    for (const Vec2f& w : s->waypoints) {
      wps.push_back(w.x);
      wps.push_back(w.y);
    }
    d.waypoints_xy = wps.data();
    d.n_waypoints = s->waypoints.size();
    d.loop_forever = s->loop_forever ? 1 : 0;
    d.agent_eyesight_range = s->agent_eyesight_range;
    keep_.push_back(s);
    const uint32_t sink = cs_mesh_add_source_sink(mesh_, &d);
    if (sink == UINT32_MAX) throw std::runtime_error(cs_mesh_last_error(mesh_));
    return sink;
  }
  void remove_source_sink(std::size_t id) { cs_mesh_remove_source_sink(mesh_, (uint32_t)id); }  // lib.rs:164
  std::size_t add_event_listener(std::shared_ptr<EventListener> l) {                           // lib.rs:171
    listeners_[next_listener_] = std::move(l);
    cs_mesh_event_recording(mesh_, 1);
    return next_listener_++;
  }
  void remove_agents(AgentId agent) {  // lib.rs:176-192
    const int rc = cs_mesh_remove_agent(mesh_, agent);
    refresh();
    if (rc != 0) throw std::runtime_error(cs_mesh_last_error(mesh_));
  }
  void step(std::chrono::duration<double> dur) {  // lib.rs:195-383
    cs_step_report rep;
    const int rc = cs_mesh_step(mesh_, dur.count(), &rep);
    refresh();
    if (rc != 0) throw std::runtime_error(cs_mesh_last_error(mesh_));
  }
  void step_no_readback(std::chrono::duration<double> dur) {  // nothing waits for the device
    if (cs_mesh_step(mesh_, dur.count(), nullptr) != 0) throw std::runtime_error(cs_mesh_last_error(mesh_));
  }
  void recut() {  // cuts to the quantiles of where the crowd stands now (in-process form)
    if (cs_mesh_recut(mesh_) != 0) throw std::runtime_error(cs_mesh_last_error(mesh_));
  }
  cs_mesh* handle() { return mesh_; }

 private:
  template <class P>
  uint32_t handle(const std::shared_ptr<P>& p) {
    auto it = handles_.find(p.get());
    if (it != handles_.end()) return it->second;
    const uint32_t h = register_planner(p.get());
    if (h == UINT32_MAX) throw std::runtime_error(cs_mesh_last_error(mesh_));
    handles_[p.get()] = h;
    keep_.push_back(p);
    return h;
  }
  uint32_t register_planner(HighLevelPlanner* p) {
    cs_hlp_desc d = p->describe();
    return cs_mesh_register_hlp(mesh_, &d);
  }
  uint32_t register_planner(LocalPlanner* p) { return p->register_with(mesh_); }
  void refresh() {
    cs_event ev[256];
    for (;;) {
      const std::size_t n = cs_mesh_drain_events(mesh_, ev, 256);
      for (std::size_t i = 0; i < n; ++i)
        for (auto& kv : listeners_) {
          if (ev[i].kind == CS_EVENT_SPAWNED) kv.second->agent_spawned({ev[i].x, ev[i].y}, ev[i].id);
          if (ev[i].kind == CS_EVENT_DESTROYED) kv.second->agent_destroyed(ev[i].id);
        }
      if (n < 256) break;
    }
    std::vector<cs_agent_view> v(cs_mesh_agent_count(mesh_));
    const std::size_t got = cs_mesh_read_agents(mesh_, v.data(), v.size());
    agents.clear();
    for (std::size_t i = 0; i < got && got != SIZE_MAX; ++i) {
      Agent a;
      a.agent_id = v[i].id;
      a.position = {v[i].x, v[i].y};
      a.velocity = {v[i].vx, v[i].vy};
      a.next_waypoint = v[i].next_waypoint;
      a.eyesight_range = v[i].eyesight_range;
      agents[a.agent_id] = a;
    }
  }

  cs_mesh* mesh_ = nullptr;
  std::map<const void*, uint32_t> handles_;
  std::vector<std::shared_ptr<void>> keep_;
  std::map<std::size_t, std::shared_ptr<EventListener>> listeners_;
  std::size_t next_listener_ = 0;
};

}  // namespace rmf_crowdsim
#endif
