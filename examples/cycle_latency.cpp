// cycle_latency.cpp — what one control cycle of the reference's own configuration costs through
// the C ABI (5 x 9 samples as reference src/sfw_planner.cpp:64-85, N pedestrians): per-call and
// whole-cycle medians of  set_costmap + set_footprint + set_agents + score_grid (blocking).
//
//   build: make -C social_force_window_planner_amd/csrc latency
//   run:   build/cycle_latency [cycles] [laser points] [markers] [map changes]
// map changes = 1: one cell of the costmap differs from cycle to cycle (the library recognises an UNCHANGED snapshot and does not
// send it again: a local costmap updates at its own rate, a few Hz, the controller hands it over at 10-20 Hz); 0 (default): the
// same map every cycle.
// markers = 1: every cycle also fetches all 45 Trajectories (what the reference's MarkerArray holds, :347-386) with
// sfw_set_points_capture on (the scoring launch leaves the points: one extra D2H); markers = 2: the same without the
// capture (the dump re-runs the rollout).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/sfw_hip.h"

using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t0) {
  return std::chrono::duration<double, std::micro>(clk::now() - t0).count();
}
static double median(std::vector<double> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main(int argc, char **argv) {
  const int cycles = argc > 1 ? std::atoi(argv[1]) : 200;
  const int n_laser = argc > 2 ? std::atoi(argv[2]) : 0;  // obstacles1: laser points near the robot
  const int markers = argc > 3 ? std::atoi(argv[3]) : 0;
  const int map_changes = argc > 4 ? std::atoi(argv[4]) : 0;
  std::vector<double> laser;
  for (int i = 0; i < n_laser; ++i) {  // a wall 1.5 m to the left and a pillar ahead
    const double u = (i + 0.5) / n_laser;
    if (i % 3) { laser.push_back(-2.0 + 5.0 * u); laser.push_back(1.5); }
    else { laser.push_back(2.5 + 0.2 * std::cos(9.0 * u)); laser.push_back(-1.0 + 0.2 * std::sin(9.0 * u)); }
  }
  const unsigned N = 200;
  const double res = 0.05, origin = -5.0;
  std::vector<uint8_t> cells(static_cast<size_t>(N) * N, 0);
  for (unsigned i = 0; i < N; ++i) cells[i] = cells[(N - 1) * N + i] = cells[i * N] = cells[i * N + N - 1] = 255;
  std::vector<double> fp;
  for (int k = 0; k < 16; ++k) {
    fp.push_back(0.35 * std::cos(k * M_PI / 8));
    fp.push_back(0.35 * std::sin(k * M_PI / 8));
  }
  const double lin[5] = {0.0, 0.175, 0.35, 0.525, 0.7};
  const double ang[9] = {0.0, 0.125, -0.125, 0.25, -0.25, 0.375, -0.375, 0.5, -0.5};
  const sfw_robot_state rs{0.0, 0.0, 0.0, 0.3, 0.0, 0.0};
  const sfw_goal_args ga{1.0, 0.0, 1.0, 2.0, 0.5};

  for (double sim_time : {1.0, 1.5})
    for (int n_people : {0, 5, 20, 50}) {
      sfw_params p;
      sfw_params_default(&p);
      p.sim_time = sim_time;
      p.sim_granularity = sim_time == 1.0 ? 0.025 : 0.25;  // code defaults / shipped yaml
      sfw_handle h = nullptr;
      if (sfw_create(&p, 0, &h) != SFW_OK) {
        std::fprintf(stderr, "sfw_create failed (no HIP device?)\n");
        return 1;
      }
      std::vector<sfw_agent> ag(1 + n_people);
      ag[0] = sfw_agent{};
      ag[0].vx = 0.3; ag[0].desired_velocity = 0.7; ag[0].radius = 0.35; ag[0].id = 0; ag[0].group_id = -1;
      for (int i = 1; i <= n_people; ++i) {
        const double a = i * 2.399963, r = 1.5 + 3.0 * i / (n_people + 1.0);
        sfw_agent q{};
        q.x = r * std::cos(a); q.y = r * std::sin(a);
        q.vx = 0.8 * std::cos(a + 2.0); q.vy = 0.8 * std::sin(a + 2.0);
        q.goal_x = q.x + 2.0 * q.vx; q.goal_y = q.y + 2.0 * q.vy;
        q.goal_radius = 0.35; q.desired_velocity = 1.0; q.radius = 0.35; q.has_goal = 1; q.id = i; q.group_id = -1;
        ag[i] = q;
      }
      std::vector<double> costs(45);
      sfw_best best;
      std::vector<double> t_map, t_fp, t_ag, t_score, t_all, t_mark;
      const int S = static_cast<int>(p.sim_time / p.sim_granularity + 0.5);
      std::vector<double> pts(static_cast<size_t>(45) * S * 3);
      std::vector<int32_t> npts(45);
      if (markers == 1) sfw_set_points_capture(h, 1);
      for (int c = 0; c < cycles + 10; ++c) {
        if (map_changes) cells[static_cast<size_t>(N) * 7 + 7 + (c % 5)] = static_cast<uint8_t>(c & 1);  // (a free cell far from the robot)
        const auto t0 = clk::now();
        int rc = sfw_set_costmap(h, cells.data(), N, N, origin, origin, res);
        const double a = us_since(t0);
        rc |= sfw_set_footprint(h, fp.data(), 16);
        const double b = us_since(t0);
        rc |= sfw_set_agents(h, ag.data(), static_cast<int>(ag.size()), n_laser ? laser.data() : nullptr, n_laser);
        const double d = us_since(t0);
        rc |= sfw_score_grid(h, &rs, lin, 5, ang, 9, &ga, costs.data(), &best);
        const double e = us_since(t0);
        if (markers) rc |= sfw_grid_points_batch(h, 0, 45, pts.data(), npts.data());
        const double f = us_since(t0);
        if (rc != SFW_OK) {
          std::fprintf(stderr, "error: %s\n", sfw_last_error(h));
          return 1;
        }
        if (c >= 10) {
          t_map.push_back(a); t_fp.push_back(b - a); t_ag.push_back(d - b); t_score.push_back(e - d); t_all.push_back(f);
          t_mark.push_back(f - e);
        }
      }
      // the scalar call sites (ref :204-206 rotate in place, :299-301): one blocking sfw_score_one with its Trajectory points
      std::vector<double> t_one;
      for (int c = 0; c < 60; ++c) {
        double cost = 0;
        int32_t np = 0;
        const auto t0 = clk::now();
        if (sfw_score_one(h, &rs, 0.0, 0.0, 0.4, &ga, &cost, pts.data(), S, &np) != SFW_OK) return 1;
        if (c >= 10) t_one.push_back(us_since(t0));
      }
      std::printf("N=%2d O=%3d S=%2d: cycle %6.1f us  (set_costmap %5.1f  set_footprint %4.1f  set_agents %5.1f  score_grid %6.1f",
                  n_people, n_laser, S, median(t_all), median(t_map), median(t_fp), median(t_ag), median(t_score));
      if (markers) std::printf("  points of 45 samples %5.1f [%s]", median(t_mark), markers == 1 ? "captured" : "re-run");
      std::printf(")  best index %lld  | score_one %5.1f us\n", static_cast<long long>(best.index), median(t_one));
      sfw_destroy(h);
    }
  return 0;
}
