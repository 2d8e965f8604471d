// multi_device.cpp — the single-process multi-device entry of the C ABI from a C++ caller: a 96 x 64 (v,w) grid with
// twelve pedestrians scored (a) by one handle, (b) by sfw_multi_* with one rank per visible device over RCCL
// (ncclCommInitAll + one ncclAllReduce(min) per call), (c) by two ranks sharing device 0 behind the host-side reduce.
// All three must agree bit for bit (costs and selection).
//
//   build: make -C social_force_window_planner_amd/csrc multidemo
//   run:   build/multi_device [n_devices]     (default: every visible device)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../include/sfw_hip.h"

namespace {
bool same(const std::vector<double> &a, const std::vector<double> &b, const sfw_best &x, const sfw_best &y) {
  return a.size() == b.size() && std::memcmp(a.data(), b.data(), sizeof(double) * a.size()) == 0 && x.index == y.index &&
         x.cost == y.cost && x.vx == y.vx && x.vtheta == y.vtheta && x.n_valid == y.n_valid;
}
}  // namespace

int main(int argc, char **argv) {
  const int N = 240, NV = 96, NW = 64, A = 13;
  std::vector<uint8_t> cells(static_cast<size_t>(N) * N, 0);
  for (int y = 150; y < 156; ++y)
    for (int x = 170; x < 176; ++x) cells[static_cast<size_t>(y) * N + x] = 254;  // one obstacle
  std::vector<double> fp(32), lin(NV), ang(NW);
  for (int i = 0; i < 16; ++i) {
    fp[2 * i] = 0.35 * std::cos(i * M_PI / 8.0);
    fp[2 * i + 1] = 0.35 * std::sin(i * M_PI / 8.0);
  }
  sfw_params params;
  sfw_params_default(&params);
  for (int i = 0; i < NV; ++i) lin[i] = i * params.max_vel_x / (NV - 1);
  for (int i = 0; i < NW / 2; ++i) {
    ang[2 * i] = (i + 0.5) * (0.5 / (NW / 2));
    ang[2 * i + 1] = -ang[2 * i];
  }
  std::vector<sfw_agent> agents(A);
  std::memset(agents.data(), 0, sizeof(sfw_agent) * A);
  agents[0].vx = 0.3;
  agents[0].desired_velocity = 0.7;
  agents[0].radius = 0.35;
  agents[0].group_id = -1;
  agents[0].id = SFW_ROBOT_ID_NONE;
  for (int i = 1; i < A; ++i) {
    sfw_agent &a = agents[i];
    const double th = 0.5 * i;
    a.x = (1.5 + 0.25 * i) * std::cos(th);
    a.y = (1.5 + 0.25 * i) * std::sin(th);
    a.vx = -0.7 * std::cos(th + 0.3);
    a.vy = -0.7 * std::sin(th + 0.3);
    a.goal_x = a.x + 2.0 * a.vx;
    a.goal_y = a.y + 2.0 * a.vy;
    a.goal_radius = a.radius = 0.35;
    a.desired_velocity = 1.0;
    a.has_goal = 1;
    a.id = i;
    a.group_id = -1;
  }
  const sfw_robot_state rs{0.0, 0.0, 0.0, 0.3, 0.0, 0.0};
  const sfw_goal_args ga{1.0, 0.0, 1.0, 2.0, 0.5};

  // (a) one handle
  sfw_handle h = nullptr;
  if (sfw_create(&params, 0, &h) != SFW_OK) {
    std::fprintf(stderr, "sfw_create failed: no HIP device (there is no CPU fallback)\n");
    return 2;
  }
  std::vector<double> c1(static_cast<size_t>(NV) * NW), c2(c1.size()), c3(c1.size());
  sfw_best b1, b2, b3;
  int rc = sfw_set_costmap(h, cells.data(), N, N, -6.0, -6.0, 0.05);
  if (rc == SFW_OK) rc = sfw_set_footprint(h, fp.data(), 16);
  if (rc == SFW_OK) rc = sfw_set_agents(h, agents.data(), A, nullptr, 0);
  if (rc == SFW_OK) rc = sfw_score_grid(h, &rs, lin.data(), NV, ang.data(), NW, &ga, c1.data(), &b1);
  if (rc != SFW_OK) {
    std::fprintf(stderr, "single: error %d: %s\n", rc, sfw_last_error(h));
    return 1;
  }
  sfw_destroy(h);

  auto run_multi = [&](const std::vector<int> &devs, int exchange, std::vector<double> &costs, sfw_best &best, double *us) {
    sfw_multi_handle m = nullptr;
    int e = sfw_multi_create(&params, devs.data(), static_cast<int32_t>(devs.size()), exchange, &m);
    if (e != SFW_OK) {
      std::fprintf(stderr, "sfw_multi_create failed: %d\n", e);
      return e;
    }
    e = sfw_multi_set_costmap(m, cells.data(), N, N, -6.0, -6.0, 0.05);
    if (e == SFW_OK) e = sfw_multi_set_footprint(m, fp.data(), 16);
    if (e == SFW_OK) e = sfw_multi_set_agents(m, agents.data(), A, nullptr, 0);
    for (int rep = 0; rep < 3 && e == SFW_OK; ++rep)  // the third call is timed warm
      e = sfw_multi_score_grid(m, &rs, lin.data(), NV, ang.data(), NW, &ga, costs.data(), &best);
    if (e != SFW_OK) std::fprintf(stderr, "multi: error %d: %s\n", e, sfw_multi_last_error(m));
    else for (int w = 0; w < 3; ++w) sfw_multi_last_us(m, w, &us[w]);
    sfw_multi_destroy(m);
    return e;
  };

  // (b) one rank per device over RCCL
  int n_dev = 0;
  (void)hipGetDeviceCount(&n_dev);
  if (argc > 1) n_dev = std::min(n_dev, std::atoi(argv[1]));
  std::vector<int> devs;
  for (int d = 0; d < n_dev; ++d) devs.push_back(d);
  double us[3];
  if (run_multi(devs, SFW_MULTI_RCCL, c2, b2, us) != SFW_OK) return 1;
  std::printf("rccl R=%d: %s (enqueue %.0f us, all-reduce + table fetch %.0f us, cost fetch %.0f us)\n", n_dev,
              same(c1, c2, b1, b2) ? "identical" : "DIFFERENT", us[0], us[1], us[2]);
  // (c) two ranks on device 0, host-side reduce
  if (run_multi({0, 0}, SFW_MULTI_HOST_REDUCE, c3, b3, us) != SFW_OK) return 1;
  std::printf("host-reduce R=2: %s (enqueue %.0f us, exchange %.0f us, cost fetch %.0f us)\n",
              same(c1, c3, b1, b3) ? "identical" : "DIFFERENT", us[0], us[1], us[2]);
  std::printf("RESULT index=%lld vx=%.4f vtheta=%.4f cost=%.6f valid=%lld\n", (long long)b1.index, b1.vx, b1.vtheta, b1.cost,
              (long long)b1.n_valid);
  return same(c1, c2, b1, b2) && same(c1, c3, b1, b3) ? 0 : 1;
}
