// drive_demo.cpp — end-to-end use of the ROS-free host classes, wired the way
// nav2's controller_server drives the reference plugin (reference
// src/sfw_planner_node.cpp:220-331): sensor callbacks -> getAgents(), global-plan
// pruning, updatePlan(), findBestAction() once per control cycle.  A differential-
// drive robot crosses a room while pedestrians walk across its path; the (v,w)
// grid of every cycle is scored on the GPU.
//
//   build: make -C social_force_window_planner_amd/host demo
//   run:   social_force_window_planner_amd/host/drive_demo [cycles] [n_people] [nv nw]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../social_force_window_planner_amd/host/plan_utils.hpp"
#include "../social_force_window_planner_amd/host/sensor_interface.hpp"
#include "../social_force_window_planner_amd/host/sfw_planner.hpp"

using namespace social_force_window_planner;

struct Walker { double x, y, vx, vy; };

int main(int argc, char **argv) {
  const int max_cycles = argc > 1 ? std::atoi(argv[1]) : 600;
  const int n_people = argc > 2 ? std::atoi(argv[2]) : 12;
  const int nv = argc > 4 ? std::atoi(argv[3]) : 0, nw = argc > 4 ? std::atoi(argv[4]) : 0;
  const double dt = 0.1;  // 10 Hz controller

  // 20 m x 20 m room, 5 cm cells, a few inflated pillars, unknown border
  const unsigned N = 400;
  const double res = 0.05, origin = -10.0;
  std::vector<uint8_t> cells(static_cast<size_t>(N) * N, 0);
  const double pillars[4][2] = {{-3.0, 2.2}, {0.5, -2.4}, {3.5, 2.6}, {-1.0, -3.5}};
  for (unsigned my = 0; my < N; ++my)
    for (unsigned mx = 0; mx < N; ++mx) {
      const double wx = origin + (mx + 0.5) * res, wy = origin + (my + 0.5) * res;
      double c = 0.0;
      for (const auto &p : pillars) {
        const double d = std::hypot(wx - p[0], wy - p[1]);
        c = std::fmax(c, d <= 0.3 ? 254.0 : std::fmax(0.0, 253.0 * (1.0 - (d - 0.3) / 0.6)));
      }
      if (mx == 0 || my == 0 || mx == N - 1 || my == N - 1) c = 255.0;
      cells[static_cast<size_t>(my) * N + mx] = static_cast<uint8_t>(c);
    }
  const CostmapView costmap{cells.data(), N, N, origin, origin, res};

  // people walking across the corridor the robot follows
  std::vector<Walker> walkers;
  unsigned lcg = 12345u;
  auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 8) * (1.0 / 16777216.0); };
  for (int i = 0; i < n_people; ++i) {
    const double x = -5.0 + 10.0 * rnd(), side = (i % 2) ? 1.0 : -1.0;
    walkers.push_back({x, side * (2.0 + 3.0 * rnd()), 0.15 * (rnd() - 0.5), -side * (0.35 + 0.5 * rnd())});
  }

  ControllerParams params;  // reference defaults (sfw_planner.hpp:56-66)
  InterfaceParams iparams;
  auto sensors = std::make_shared<SFMSensorInterface>(iparams);
  std::vector<Point> footprint;
  for (int k = 0; k < 16; ++k) footprint.push_back(Point{0.35 * std::cos(k * M_PI / 8), 0.35 * std::sin(k * M_PI / 8), 0.0});
  SFWPlanner planner(params, sensors, costmap, footprint, /*device*/ 0);
  if (nv > 0 && nw > 0) {  // a denser window than the reference's 5 x 9
    std::vector<double> lin, ang;
    for (int i = 0; i < nv; ++i) lin.push_back(nv > 1 ? i * params.max_vel_x_ / (nv - 1) : params.max_vel_x_);
    const int half = nw / 2;
    ang.push_back(0.0);
    for (int i = 1; i <= half; ++i) { ang.push_back(i * params.max_vel_th_ / half); ang.push_back(-i * params.max_vel_th_ / half); }
    planner.setSampleSets(lin, ang);
  }

  // global plan: straight line, a pose every 0.25 m
  std::vector<PoseStamped> global_plan;
  for (double x = -7.0; x <= 7.0 + 1e-9; x += 0.25) {
    PoseStamped p;
    p.frame_id = "odom";
    p.pose.position.x = x;
    global_plan.push_back(p);
  }
  sensors->start();  // setPlan() does this in the node (ref :114-117)

  double rx = -7.0, ry = 0.0, rth = 0.0, rv = 0.0, rw = 0.0, min_clear = 1e9;
  int cycle = 0, failed = 0;
  bool reached = false;
  for (; cycle < max_cycles && !reached; ++cycle) {
    // --- sensor callbacks (odometry first: the interface ignores the rest until it has one)
    Odometry od;
    od.frame_id = "odom";
    od.pose.position.x = rx;
    od.pose.position.y = ry;
    od.pose.orientation = quaternionFromYaw(rth);
    od.twist.linear.x = rv;
    od.twist.angular.z = rw;
    sensors->odomCb(od);
    People pp;
    pp.frame_id = "odom";
    for (size_t i = 0; i < walkers.size(); ++i) {
      Person p;
      p.position = Point{walkers[i].x, walkers[i].y, std::atan2(walkers[i].vy, walkers[i].vx)};
      p.velocity = Vector3{walkers[i].vx, walkers[i].vy, 0.0};
      p.tags = {std::to_string(i + 1), "-1"};
      pp.people.push_back(p);
    }
    sensors->peopleCb(pp);

    // --- computeVelocityCommands (ref :220-331)
    PoseStamped pose;
    pose.frame_id = "odom";
    pose.pose.position.x = rx;
    pose.pose.position.y = ry;
    pose.pose.orientation = quaternionFromYaw(rth);
    std::vector<PoseStamped> local;
    try {
      local = transformGlobalPlan(global_plan, pose, N, N, res, [](const PoseStamped &in, PoseStamped &out) { out = in; return true; });
    } catch (const PlannerException &e) {
      std::printf("cycle %d: %s\n", cycle, e.what());
      break;
    }
    planner.updatePlan(local);
    Twist vel, cmd;
    vel.linear.x = rv;
    vel.angular.z = rw;
    const bool ok = planner.findBestAction(pose, vel, cmd);
    if (!ok) ++failed;
    reached = planner.isGoalReached();

    // --- the world moves on
    rv = cmd.linear.x;
    rw = cmd.angular.z;
    rx += rv * std::cos(rth) * dt;
    ry += rv * std::sin(rth) * dt;
    rth += rw * dt;
    for (auto &w : walkers) {
      w.x += w.vx * dt;
      w.y += w.vy * dt;
      if (std::fabs(w.y) > 6.0) w.vy = -w.vy;  // turn around at the walls
      min_clear = std::fmin(min_clear, std::hypot(w.x - rx, w.y - ry));
    }
    if (cycle % 25 == 0)
      std::printf("cycle %4d  pose (%6.2f, %5.2f, %5.2f)  cmd (%.3f, %+.3f)  branch %d  valid %lld/%zu\n", cycle, rx, ry, rth,
                  rv, rw, planner.lastBranch(), static_cast<long long>(planner.lastBest().n_valid),
                  planner.lastCosts().size());
  }
  const double dist_goal = std::hypot(rx - 7.0, ry - 0.0);
  std::printf("RESULT cycles=%d reached=%d dist_goal=%.3f min_clearance=%.3f failed_cycles=%d\n", cycle, reached ? 1 : 0,
              dist_goal, min_clear, failed);
  return reached ? 0 : 1;
}
