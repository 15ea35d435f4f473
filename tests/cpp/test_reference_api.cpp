// The reference's two step tests, restated over the C++ mirror (include/crowdsim.hpp):
//   test_step_integration                 rmf_crowdsim/src/lib.rs:423-453
//   test_event_listener_source_sink_api   rmf_crowdsim/tests/event_listeners_test.rs:65-111
// Runs on an MI355X (tests/test_gpu_cpp_api.py builds and launches it).
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "crowdsim.hpp"

using namespace rmf_crowdsim;

#define CHECK(cond)                                                      \
  do {                                                                   \
    if (!(cond)) {                                                       \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);      \
      std::exit(1);                                                      \
    }                                                                    \
  } while (0)

struct MockEventListener : EventListener {
  std::vector<AgentId> added, removed;
  void agent_spawned(Vec2f, AgentId agent) override { added.push_back(agent); }
  void agent_destroyed(AgentId agent) override { removed.push_back(agent); }
};

static void test_step_integration() {
  Vec2f velocity{1.0, 0.0};
  auto step_size = std::chrono::duration<double>(1.0);
  Simulation crowd_simulation(LocationHash2D(1000.0, 1000.0, 20.0, Point{-500.0, -500.0}));
  CHECK(crowd_simulation.agents.size() == 0);
  auto agents = crowd_simulation.add_agents({Point{0.0, 0.0}}, std::make_shared<StubHighLevelPlan>(velocity),
                                            std::make_shared<NoLocalPlan>(), 100.0);
  CHECK(agents.size() == 1);
  CHECK(crowd_simulation.agents.size() == 1);
  crowd_simulation.step(step_size);
  CHECK(crowd_simulation.agents.size() == 1);
  const Agent& a = crowd_simulation.agents.at(0);
  CHECK(std::hypot(a.position.x - velocity.x, a.position.y - velocity.y) < 1e-5);
}

static void test_event_listener_source_sink_api() {
  auto step_size = std::chrono::duration<double>(1.0);
  Simulation crowd_simulation(LocationHash2D(1000.0, 1000.0, 20.0, Point{-500.0, -500.0}));
  auto source_sink = std::make_shared<SourceSink>();
  source_sink->source = {0.0, 0.0};
  source_sink->waypoints = {Vec2f{20.0, 0.0}};
  source_sink->radius_sink = 1.0;
  source_sink->crowd_generator = std::make_shared<MonotonicCrowd>(1.0);
  source_sink->high_level_planner = std::make_shared<StubHighLevelPlan>(Vec2f{1.0, 0.0});
  source_sink->local_planner = std::make_shared<NoLocalPlan>();
  source_sink->agent_eyesight_range = 5.0;
  source_sink->loop_forever = false;
  auto event_listener = std::make_shared<MockEventListener>();
  crowd_simulation.add_event_listener(event_listener);
  crowd_simulation.add_source_sink(source_sink);
  for (std::size_t steps = 0; steps < 20; ++steps) {
    CHECK(crowd_simulation.agents.size() == steps);
    CHECK(event_listener->added.size() == steps);
    crowd_simulation.step(step_size);
  }
  for (std::size_t steps = 20; steps < 40; ++steps) {
    CHECK(crowd_simulation.agents.size() == 20);
    CHECK(event_listener->added.size() == steps);
    CHECK(event_listener->removed.size() == steps - 20);
    crowd_simulation.step(step_size);
  }
}

static void test_index_out_of_bounds_is_an_error() {
  Simulation sim(LocationHash2D(2.0, 2.0, 1.0, Point{0.0, 0.0}));
  bool threw = false;
  try {
    sim.add_agents({Point{5.0, 0.5}}, std::make_shared<StubHighLevelPlan>(Vec2f{0, 0}),
                   std::make_shared<NoLocalPlan>(), 1.0);
  } catch (const std::runtime_error& e) {
    threw = std::string(e.what()) == "Index out of bounds";
  }
  CHECK(threw);
}

static void test_viz_scene() {  // rmf_crowdsim_viz/src/main.rs:64-94
  Simulation sim(LocationHash2D(1000.0, 1000.0, 20.0, Point{-500.0, -500.0}));
  sim.add_agents({Point{100, 100}, Point{100, -100}, Point{60, 100}},
                 std::make_shared<IdParityHighLevelPlan>(Vec2f{0.0, 10.0}),
                 std::make_shared<Zanlungo>(1.0, 1.0, 0.0, 40.0, 2.0, 20.0), 100.0);
  for (int k = 0; k < 300; ++k) sim.step(std::chrono::duration<double>(0.05));
  CHECK(std::fabs(sim.agents.at(0).position.x - 100.0) > 5.0);  // agent 0 dodged agent 1
  CHECK(std::isfinite(sim.agents.at(1).position.y));
}

// streaming view: frames requested behind steps that nobody waits for
static void test_snapshots() {
  Simulation sim(LocationHash2D(100.0, 100.0, 2.0, Point{-50.0, -50.0}));
  auto hlp = std::make_shared<StubHighLevelPlan>(Vec2f{1.0, 0.0});
  auto lp = std::make_shared<NoLocalPlan>();
  sim.add_agents({Point{0.0, 0.0}, Point{3.0, 4.0}}, hlp, lp, 2.0);
  CHECK(!sim.snapshot().ready);
  for (int k = 0; k < 10; ++k) {
    sim.step_no_readback(std::chrono::duration<double>(0.1));
    sim.request_snapshot();
  }
  Simulation::Frame f = sim.snapshot();
  CHECK(f.ready && f.count == 2 && f.step_index == 10);
  for (std::size_t i = 0; i < f.count; ++i) {
    const double x0 = f.agents[i].id == 0 ? 0.0 : 3.0;
    CHECK(std::fabs(f.agents[i].x - (x0 + 1.0)) < 1e-5 && std::fabs(f.agents[i].vx - 1.0f) < 1e-6);
  }
}

// A LocalPlanner written by the user (local_planner.rs:7-18): host code, evaluated through the engine's batched
// callback.  This one steers away from the mean position of whoever it sees; with one neighbour 1 m to the right
// the answer is known in closed form.
struct ShyPlanner : LocalPlanner {
  mutable int calls = 0;
  Vec2f get_desired_velocity(const Agent& agent, const std::vector<Agent>& nearby, Vec2f recommended) const override {
    ++calls;
    Vec2f w = recommended;
    for (const Agent& other : nearby) {
      w.x -= 0.5 * (other.position.x - agent.position.x);
      w.y -= 0.5 * (other.position.y - agent.position.y);
    }
    return w;
  }
};

static void test_user_local_planner() {
  Simulation sim(LocationHash2D(100.0, 100.0, 2.0, Point{0.0, 0.0}));
  auto shy = std::make_shared<ShyPlanner>();
  sim.add_agents({Point{10.0, 10.0}, Point{11.0, 10.0}, Point{50.0, 50.0}}, std::make_shared<StubHighLevelPlan>(Vec2f{0.0, 1.0}),
                 shy, 3.0);
  sim.step(std::chrono::duration<double>(0.1));
  CHECK(shy->calls == 3);
  // agent 0 sees agent 1 at +1 m in x: w = (0, 1) - 0.5 * (1, 0); agent 1 the mirror image; agent 2 sees nobody
  CHECK(std::hypot(sim.agents.at(0).position.x - (10.0 - 0.05), sim.agents.at(0).position.y - 10.1) < 1e-5);
  CHECK(std::hypot(sim.agents.at(1).position.x - (11.0 + 0.05), sim.agents.at(1).position.y - 10.1) < 1e-5);
  CHECK(std::hypot(sim.agents.at(2).position.x - 50.0, sim.agents.at(2).position.y - 50.1) < 1e-5);
}

// The reference's source-sink test on a 2 x 2 tile mesh (TiledSimulation over cs_mesh_*): same populations, same events
static void test_event_listener_source_sink_api_on_a_mesh() {
  TiledSimulation sim(LocationHash2D(1000.0, 1000.0, 20.0, Point{-500.0, -500.0}), 2, 2, 1);
  auto listener = std::make_shared<MockEventListener>();
  sim.add_event_listener(listener);
  auto ss = std::make_shared<SourceSink>();
  ss->source = Vec2f{-10.0, 0.0};           // the lane crosses the cut at x = 0
  ss->waypoints = {Vec2f{10.0, 0.0}};
  ss->radius_sink = 1.0;
  ss->crowd_generator = std::make_shared<MonotonicCrowd>(1.0);
  ss->high_level_planner = std::make_shared<StubHighLevelPlan>(Vec2f{1.0, 0.0});
  ss->local_planner = std::make_shared<NoLocalPlan>();
  ss->agent_eyesight_range = 5.0;
  ss->loop_forever = false;
  sim.add_source_sink(ss);
  for (std::size_t steps = 0; steps < 40; ++steps) {
    CHECK(sim.agents.size() == (steps < 20 ? steps : 20));
    CHECK(listener->added.size() == steps);
    CHECK(listener->removed.size() == (steps < 20 ? 0 : steps - 20));
    sim.step(std::chrono::duration<double>(1.0));
  }
}

// PoissonCrowd (source_sink.rs:63-82): unseeded, so only its statistics can be checked.  dt * rate = 0.5: a sink whose
// source is always free (the walker leaves 1 m per step) spawns in a step with probability 1 - e^-0.5 = 0.393, one agent
// at most (lib.rs:207-219).
static void test_poisson_crowd() {
  Simulation sim(LocationHash2D(1000.0, 1000.0, 20.0, Point{-500.0, -500.0}));
  auto ss = std::make_shared<SourceSink>();
  ss->source = Vec2f{0.0, 0.0};
  ss->waypoints = {Vec2f{400.0, 0.0}};
  ss->radius_sink = 1.0;
  ss->crowd_generator = std::make_shared<PoissonCrowd>(0.5);
  ss->high_level_planner = std::make_shared<StubHighLevelPlan>(Vec2f{1.0, 0.0});
  ss->local_planner = std::make_shared<NoLocalPlan>();
  ss->agent_eyesight_range = 5.0;
  ss->loop_forever = false;
  sim.add_source_sink(ss);
  for (int k = 0; k < 300; ++k) sim.step(std::chrono::duration<double>(1.0));
  const double spawned = (double)sim.agents.size();  // nobody reaches the sink within 300 m
  CHECK(spawned > 300 * 0.393 - 5 * 8.5 && spawned < 300 * 0.393 + 5 * 8.5);  // +- 5 sigma
  PoissonCrowd g(40.0);
  double sum = 0;
  for (int k = 0; k < 4000; ++k) sum += (double)g.get_number_to_spawn(std::chrono::duration<double>(0.05));
  CHECK(std::fabs(sum / 4000 - 2.0) < 0.15);
}

int main() {
  test_poisson_crowd();
  test_event_listener_source_sink_api_on_a_mesh();
  test_user_local_planner();
  test_snapshots();
  test_step_integration();
  test_event_listener_source_sink_api();
  test_index_out_of_bounds_is_an_error();
  test_viz_scene();
  std::printf("8 passed\n");
  return 0;
}
