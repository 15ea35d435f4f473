// A 2 x 2 in-process tile mesh driven through the C ABI's cs_mesh_* calls from C++ (no Python in the way), against
// ONE engine on the same crowd: same agents, bit for bit; source-sinks, a re-cut and merged queries included.
// Runs on an MI355X (tests/test_gpu_cpp_api.py builds and launches it).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "crowdstep.h"

#define CHECK(cond)                                                 \
  do {                                                              \
    if (!(cond)) {                                                  \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                 \
    }                                                               \
  } while (0)

// splitmix64: the scene is generated here, not read from anywhere
static double uniform01(unsigned long long& s) {
  unsigned long long z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (double)((z ^ (z >> 31)) >> 11) / 9007199254740992.0;
}

int main() {
  cs_grid_desc grid{120.0, 120.0, 2.0, 0.0, 0.0};
  cs_mesh_desc md;
  std::memset(&md, 0, sizeof md);
  md.tiles_x = 2;
  md.tiles_y = 2;
  md.halo_cells = 1;
  md.n_ranks = 1;
  md.density_per_cell = 12.0;
  cs_mesh* mesh = cs_mesh_create(&grid, &md);
  CHECK(mesh && cs_mesh_local_tiles(mesh) == 4);
  cs_engine* one = cs_create(&grid, nullptr);
  CHECK(one);

  // planners: the same handles on the mesh and on the single engine (same registration order)
  cs_zanlungo_params z{1.0, 1.0, 0.0, 0.4, 2.0, 0.2};
  const uint32_t lp_m = cs_mesh_register_zanlungo(mesh, &z), lp_1 = cs_register_zanlungo(one, &z);
  cs_hlp_desc up, down;
  std::memset(&up, 0, sizeof up);
  up.kind = CS_HLP_CONSTANT;
  up.vx = 0.65;  // everybody drifts over the cuts at walking speed; the two flows creep against each other (a head-on
  up.vy = 0.651; // counter-flow at walking speed blows the reference's model up: DESIGN.md section 5)
  down = up;
  down.vy = 0.649;
  const uint32_t up_m = cs_mesh_register_hlp(mesh, &up), up_1 = cs_register_hlp(one, &up);
  const uint32_t dn_m = cs_mesh_register_hlp(mesh, &down), dn_1 = cs_register_hlp(one, &down);
  CHECK(lp_m == lp_1 && up_m == up_1 && dn_m == dn_1);

  // a jittered lattice of 9,025 agents in the middle of the grid (two interleaved groups, drifting over every cut)
  std::vector<double> a_xy, b_xy;
  unsigned long long seed = 42;
  for (int ix = 0; ix < 95; ++ix)
    for (int iy = 0; iy < 95; ++iy) {
      const double x = 30.0 + 0.63 * ix + 0.1 * (uniform01(seed) - 0.5), y = 30.0 + 0.63 * iy + 0.1 * (uniform01(seed) - 0.5);
      std::vector<double>& dst = ((ix + iy) & 1) ? a_xy : b_xy;
      dst.push_back(x);
      dst.push_back(y);
    }
  std::vector<uint64_t> ids_m(a_xy.size() / 2), ids_1(a_xy.size() / 2);
  CHECK(cs_mesh_add_agents(mesh, a_xy.data(), a_xy.size() / 2, up_m, lp_m, 2.0, ids_m.data()) == 0);
  CHECK(cs_add_agents(one, a_xy.data(), a_xy.size() / 2, up_1, lp_1, 2.0, ids_1.data()) == 0);
  CHECK(ids_m == ids_1);
  ids_m.resize(b_xy.size() / 2);
  CHECK(cs_mesh_add_agents(mesh, b_xy.data(), b_xy.size() / 2, dn_m, lp_m, 2.0, ids_m.data()) == 0);
  CHECK(cs_add_agents(one, b_xy.data(), b_xy.size() / 2, dn_1, lp_1, 2.0, nullptr) == 0);

  // a source-sink lane across the horizontal cut (seeded Poisson generator: the same draws on every tile)
  double wp[2] = {100.0, 62.0};
  cs_source_sink_desc sd;
  std::memset(&sd, 0, sizeof sd);
  sd.source_x = 100.0;
  sd.source_y = 40.0;
  sd.radius_sink = 0.8;
  sd.generator_kind = CS_GEN_POISSON_SEEDED;
  sd.rate = 6.0;
  sd.seed = 7;
  sd.lp = lp_m;
  cs_hlp_desc north = up;
  north.vx = 0.0;
  north.vy = 1.3;
  sd.hlp = cs_mesh_register_hlp(mesh, &north);
  CHECK(sd.hlp == cs_register_hlp(one, &north));
  sd.waypoints_xy = wp;
  sd.n_waypoints = 1;
  sd.agent_eyesight_range = 2.0;
  CHECK(cs_mesh_add_source_sink(mesh, &sd) == cs_add_source_sink(one, &sd));

  auto same = [&](const char* when) {
    const size_t n = cs_agent_count(one), n_mesh = cs_mesh_agent_count(mesh);
    if (n_mesh != n) std::printf("%s: %zu agents on the mesh, %zu on the engine (%s | %s)\n", when, n_mesh, n, cs_mesh_last_error(mesh), cs_last_error(one));
    CHECK(n_mesh == n);
    std::vector<cs_agent_view> a(n), b(n);
    CHECK(cs_read_agents(one, a.data(), n) == n && cs_mesh_read_agents(mesh, b.data(), n) == n);
    CHECK(std::memcmp(a.data(), b.data(), n * sizeof(cs_agent_view)) == 0);
    return n;
  };
  for (int k = 0; k < 150; ++k) {
    cs_step_report r1, rm;
    const bool report = k == 70 || k == 71;
    if (cs_step(one, 0.05, report ? &r1 : nullptr) != 0) std::printf("step %d, engine: %s\n", k, cs_last_error(one));
    if (cs_mesh_step(mesh, 0.05, report ? &rm : nullptr) != 0) std::printf("step %d, mesh: %s\n", k, cs_mesh_last_error(mesh));
    CHECK(cs_synchronize(one) == 0 && cs_mesh_synchronize(mesh) == 0);
    if (report) CHECK(r1.n_agents == rm.n_agents && r1.n_spawned == rm.n_spawned && r1.n_destroyed == rm.n_destroyed);
  }
  const size_t n = same("after 150 steps");
  CHECK(n > 9025);  // the lane has spawned

  // re-cut (the state must not change), then merged queries against the single engine's
  uint64_t counts[4];
  CHECK(cs_mesh_recut(mesh) == 0 && cs_mesh_tile_counts(mesh, counts) == 0);
  CHECK(counts[0] + counts[1] + counts[2] + counts[3] == n);
  same("after the re-cut");
  const double q[4] = {60.0, 60.0, 45.0, 75.5}, rad[2] = {3.0, 1.5};
  uint64_t ids_a[2 * 256], ids_b[2 * 256], cnt_a[2], cnt_b[2];
  CHECK(cs_query_radius_batch(one, 2, q, rad, 256, ids_a, cnt_a, nullptr, nullptr) == 0);
  CHECK(cs_mesh_query_radius_batch(mesh, 2, q, rad, 256, ids_b, cnt_b) == 0);
  for (int k = 0; k < 2; ++k) {
    CHECK(cnt_a[k] == cnt_b[k] && cnt_a[k] > 5 && cnt_a[k] <= 256);
    CHECK(std::memcmp(ids_a + 256 * k, ids_b + 256 * k, cnt_a[k] * sizeof(uint64_t)) == 0);
  }
  CHECK(cs_query_knn_batch(one, 2, q, 5, ids_a, cnt_a, nullptr) == 0 && cs_mesh_query_knn_batch(mesh, 2, q, 5, ids_b, cnt_b) == 0);
  CHECK(cnt_a[0] == 5 && cnt_b[0] == 5 && std::memcmp(ids_a, ids_b, 10 * sizeof(uint64_t)) == 0);

  // removal, then on
  CHECK(cs_remove_agent(one, 17) == 0 && cs_mesh_remove_agent(mesh, 17) == 0);
  CHECK(cs_mesh_remove_agent(mesh, 17) == 2);
  for (int k = 0; k < 60; ++k) CHECK(cs_step(one, 0.05, nullptr) == 0 && cs_mesh_step(mesh, 0.05, nullptr) == 0);
  same("60 steps after the removal");
  cs_mesh_destroy(mesh);
  cs_destroy(one);

  // the distributed form over a transport the HOST brings, with the one rank a single process has: a loop-back
  // transport (nobody to exchange with, an all-reduce of one, an all-gather of one) drives the same code the ranks of
  // an MPI host would (tests/test_native_mesh.py runs it with two ranks over gloo)
  static int calls[3] = {0, 0, 0};
  cs_mesh_host_transport loop;
  loop.user = calls;
  loop.exchange = [](void* u, size_t n_msgs, const int32_t*, const int32_t*, const int32_t*, const void* const*, void* const*,
                     const size_t*) -> int {
    ++static_cast<int*>(u)[0];
    return n_msgs == 0 ? 0 : 1;  // a 1 x 1 mesh has no neighbours
  };
  loop.allreduce_max_i32 = [](void* u, int32_t*, size_t) -> int {
    ++static_cast<int*>(u)[1];
    return 0;
  };
  loop.allgather = [](void* u, const void* mine, size_t bytes, void* all) -> int {
    ++static_cast<int*>(u)[2];
    std::memcpy(all, mine, bytes);
    return 0;
  };
  cs_mesh_desc ld;
  std::memset(&ld, 0, sizeof ld);
  ld.tiles_x = ld.tiles_y = 1;
  ld.halo_cells = 1;
  ld.n_ranks = 1;
  ld.host_transport = &loop;
  cs_mesh* solo = cs_mesh_create(&grid, &ld);
  CHECK(solo && cs_mesh_local_tiles(solo) == 1 && calls[1] == 1);  // (the layout checksum went through the all-reduce)
  cs_engine* ref = cs_create(&grid, nullptr);
  const uint32_t lp_s = cs_mesh_register_zanlungo(solo, &z), up_s = cs_mesh_register_hlp(solo, &up);
  CHECK(lp_s == cs_register_zanlungo(ref, &z) && up_s == cs_register_hlp(ref, &up));
  CHECK(cs_mesh_add_agents(solo, a_xy.data(), a_xy.size() / 2, up_s, lp_s, 2.0, nullptr) == 0);
  CHECK(cs_add_agents(ref, a_xy.data(), a_xy.size() / 2, up_s, lp_s, 2.0, nullptr) == 0);
  sd.lp = lp_s;
  sd.hlp = cs_mesh_register_hlp(solo, &north);
  CHECK(sd.hlp == cs_register_hlp(ref, &north));
  CHECK(cs_mesh_add_source_sink(solo, &sd) == cs_add_source_sink(ref, &sd));
  for (int k = 0; k < 80; ++k) CHECK(cs_step(ref, 0.05, nullptr) == 0 && cs_mesh_step(solo, 0.05, nullptr) == 0);
  CHECK(cs_mesh_recut(solo) == 0);
  const size_t n_solo = cs_agent_count(ref);
  CHECK(cs_mesh_agent_count(solo) == n_solo && n_solo > a_xy.size() / 2);
  std::vector<cs_agent_view> va(n_solo), vb(n_solo);
  CHECK(cs_read_agents(ref, va.data(), n_solo) == n_solo && cs_mesh_read_agents(solo, vb.data(), n_solo) == n_solo);
  CHECK(std::memcmp(va.data(), vb.data(), n_solo * sizeof(cs_agent_view)) == 0);
  CHECK(cs_query_radius_batch(ref, 2, q, rad, 256, ids_a, cnt_a, nullptr, nullptr) == 0);
  CHECK(cs_mesh_query_radius_batch(solo, 2, q, rad, 256, ids_b, cnt_b) == 0);
  CHECK(cnt_a[0] == cnt_b[0] && std::memcmp(ids_a, ids_b, cnt_a[0] * sizeof(uint64_t)) == 0);
  CHECK(calls[0] == 80 && calls[1] > 80 && calls[2] > 4);  // every step exchanged and OR-ed its spawn flags through the host
  cs_mesh_destroy(solo);
  cs_destroy(ref);
  std::printf("mesh api: passed (%zu agents)\n", n);
  return 0;
}
