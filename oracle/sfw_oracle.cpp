// sfw_oracle.cpp — CPU restatement of the reference's DWA rollout + social-force
// scoring path.  TEST INFRASTRUCTURE ONLY: nothing in the product path
// (social_force_window_planner_amd/, include/) may include, link or call this
// file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// use it, and only as the checker / the timed CPU baseline.
//
// PARITY STATUS
//   * Bresenham cell sequences: pinned against the reference's own
//     line_iterator.hpp (oracle/_ref, tests/test_oracle_kat.py).
//   * scoreTrajectory / computeSocialWork / selection rule / footprint cost:
//     restated from the reference text (citations below); the reference has no
//     tests or golden vectors and its .cpp files cannot be compiled here
//     (32 missing ROS2/nav2/lightsfm headers), so these are checked by
//     closed-form cases only.
//   * lightsfm (robotics-upo/lightsfm, header-only, NO pinned version,
//     reference package.xml:31) and nav2_costmap_2d::Costmap2D (ROS 2 Foxy) are
//     absent from /root/reference and from this image.  Their arithmetic is
//     restated from the published model (Moussaid et al. 2009/2010 social
//     force with velocity/angle interaction terms) and the call sites in
//     src/sfw_planner.cpp:592,594,697.  => PARITY UNPINNED at those two
//     boundaries.
//
// All arithmetic is double except where the reference itself uses float
// (normalizeAngle, robot_radius_).  "ref:" comments cite /root/reference.

#include "../include/sfw_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace sfwo {

// ---------------------------------------------------------------------------
// utils::Vector2d / utils::Angle subset (lightsfm vector2d.hpp / angle.hpp)
// ---------------------------------------------------------------------------
struct V2 {
  double x = 0.0, y = 0.0;
};
static inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
static inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
static inline V2 operator*(double s, V2 a) { return {s * a.x, s * a.y}; }
static inline V2 operator*(V2 a, double s) { return {a.x * s, a.y * s}; }
static inline V2 operator/(V2 a, double s) { return {a.x / s, a.y / s}; }
static inline V2 operator-(V2 a) { return {-a.x, -a.y}; }
static inline double norm(V2 a) { return std::sqrt(a.x * a.x + a.y * a.y); }
// Vector2d::normalized(): leaves a zero vector untouched.
static inline V2 normalized(V2 a) {
  double n = norm(a);
  if (n > 0.0) return {a.x / n, a.y / n};
  return a;
}
static inline V2 left_normal(V2 a) { return {-a.y, a.x}; }
// utils::Angle keeps its value in (-pi, pi].
static inline double wrap_angle(double v) {
  while (v <= -M_PI) v += 2.0 * M_PI;
  while (v > M_PI) v -= 2.0 * M_PI;
  return v;
}
static inline double angle_of(V2 a) { return wrap_angle(std::atan2(a.y, a.x)); }
static inline int angle_sign(double v) { return v == 0.0 ? 0 : (v > 0.0 ? 1 : -1); }

// ---------------------------------------------------------------------------
// sfm::Agent subset (fields the hot path reads)
// ---------------------------------------------------------------------------
struct Goal {
  V2 center;
  double radius = 0.0;
};
struct Agent {
  V2 position, velocity;
  double yaw = 0.0;
  double desiredVelocity = 0.6, radius = 0.35;
  std::vector<Goal> goals;  // std::list in lightsfm; front() = [0]
  bool cyclicGoals = false, teleoperated = false;
  double linearVelocity = 0.0, angularVelocity = 0.0;
  int groupId = -1, id = 0;
  V2 desiredForce, obstacleForce, socialForce, globalForce;
  const std::vector<V2> *obstacles1 = nullptr;  // shared laser points
};

struct World {
  sfw_params p;
  // Costmap2D stand-in
  std::vector<uint8_t> cells;
  uint32_t size_x = 0, size_y = 0;
  double origin_x = 0, origin_y = 0, resolution = 1;
  std::vector<V2> footprint;
  std::vector<Agent> agents;
  std::vector<V2> obstacles;
  std::string err;

  // nav2_costmap_2d::Costmap2D::worldToMap (Foxy): reject below origin,
  // truncate, then bound check.
  bool worldToMap(double wx, double wy, unsigned &mx, unsigned &my) const {
    if (wx < origin_x || wy < origin_y) return false;
    mx = static_cast<unsigned>((wx - origin_x) / resolution);
    my = static_cast<unsigned>((wy - origin_y) / resolution);
    return mx < size_x && my < size_y;
  }
  uint8_t getCost(unsigned mx, unsigned my) const {
    return cells[static_cast<size_t>(my) * size_x + mx];
  }
};

// ---------------------------------------------------------------------------
// Bresenham walk.  ref: include/social_force_window_planner/line_iterator.hpp:39-97
// Emits max(|dx|,|dy|)+1 cells starting at (x0,y0).
// ---------------------------------------------------------------------------
template <class F>
static inline void bresenham(int x0, int y0, int x1, int y1, F &&visit) {
  const int adx = std::abs(x1 - x0), ady = std::abs(y1 - y0);
  const int sx = (x1 >= x0) ? 1 : -1, sy = (y1 >= y0) ? 1 : -1;
  const bool x_major = adx >= ady;
  const int den = x_major ? adx : ady;
  const int add = x_major ? ady : adx;
  int num = den / 2, x = x0, y = y0;
  for (int n = 0; n <= den; ++n) {
    if (!visit(x, y)) return;
    num += add;
    if (num >= den) {
      num -= den;
      if (x_major) y += sy; else x += sx;   // minor axis step
    }
    if (x_major) x += sx; else y += sy;     // major axis step
  }
}

// ref: src/costmap_model.cpp:112-121  (253 is NOT rejected on edges)
static inline double point_cost(const World &w, int x, int y) {
  uint8_t c = w.getCost(static_cast<unsigned>(x), static_cast<unsigned>(y));
  if (c == 255) return -2.0;
  if (c == 254) return -1.0;
  return c;
}
// ref: src/costmap_model.cpp:95-110
static double line_cost(const World &w, int x0, int x1, int y0, int y1) {
  double best = 0.0, bad = 0.0;
  bool hit = false;
  bresenham(x0, y0, x1, y1, [&](int x, int y) {
    double pc = point_cost(w, x, y);
    if (pc < 0) { bad = pc; hit = true; return false; }
    if (best < pc) best = pc;
    return true;
  });
  return hit ? bad : best;
}
// ref: src/costmap_model.cpp:21-92
static double footprint_cost_oriented(const World &w, V2 pos, const std::vector<V2> &fp) {
  unsigned cx, cy;
  if (!w.worldToMap(pos.x, pos.y, cx, cy)) return -3.0;
  if (fp.size() < 3) {
    uint8_t c = w.getCost(cx, cy);
    if (c == 255) return -2.0;
    if (c == 254 || c == 253) return -1.0;
    return c;
  }
  double fc = 0.0;
  const size_t K = fp.size();
  for (size_t e = 0; e < K; ++e) {
    // edges 0..K-2 join fp[e]->fp[e+1]; the closing edge joins back()->front()
    const V2 a = (e + 1 < K) ? fp[e] : fp[K - 1];
    const V2 b = (e + 1 < K) ? fp[e + 1] : fp[0];
    unsigned x0, y0, x1, y1;
    if (!w.worldToMap(a.x, a.y, x0, y0)) return -3.0;
    if (!w.worldToMap(b.x, b.y, x1, y1)) return -3.0;
    double lc = line_cost(w, (int)x0, (int)x1, (int)y0, (int)y1);
    fc = std::max(lc, fc);
    if (lc < 0) return lc;
  }
  return fc;
}
// ref: include/social_force_window_planner/world_model.hpp:45-75
static double footprint_cost(const World &w, double x, double y, double th) {
  const double c = std::cos(th), s = std::sin(th);
  std::vector<V2> oriented;
  oriented.reserve(w.footprint.size());
  for (const V2 &q : w.footprint)
    oriented.push_back({x + (q.x * c - q.y * s), y + (q.x * s + q.y * c)});
  return footprint_cost_oriented(w, {x, y}, oriented);
}

// ---------------------------------------------------------------------------
// lightsfm subset (UNPINNED restatement)
// ---------------------------------------------------------------------------
static V2 sfm_desired(const sfw_params &p, Agent &a) {
  V2 dir;
  if (!a.goals.empty() && norm(a.goals.front().center - a.position) > a.goals.front().radius) {
    dir = normalized(a.goals.front().center - a.position);
    a.desiredForce = p.sfm_force_factor_desired * (dir * a.desiredVelocity - a.velocity) /
                     p.sfm_relaxation_time;
  } else {
    a.desiredForce = -a.velocity / p.sfm_relaxation_time;
  }
  return dir;
}
static void sfm_obstacle(const sfw_params &p, Agent &a) {
  a.obstacleForce = {0, 0};
  if (a.obstacles1 && !a.obstacles1->empty()) {
    for (const V2 &o : *a.obstacles1) {
      V2 md = a.position - o;
      double dist = norm(md) - a.radius;
      a.obstacleForce = a.obstacleForce + p.sfm_force_factor_obstacle *
                                              std::exp(-dist / p.sfm_force_sigma_obstacle) *
                                              normalized(md);
    }
    a.obstacleForce = a.obstacleForce / (double)a.obstacles1->size();
  }
}
// Force exerted on `me` by `other` (one term of computeSocialForce).
static inline V2 sfm_pair(const sfw_params &p, const Agent &me, const Agent &other) {
  V2 diff = other.position - me.position;
  V2 diffDir = normalized(diff);
  V2 velDiff = me.velocity - other.velocity;
  V2 inter = p.sfm_lambda * velDiff + diffDir;
  double interLen = norm(inter);
  V2 interDir = inter / interLen;
  double theta = wrap_angle(angle_of(diffDir) - angle_of(interDir));
  double B = p.sfm_gamma * interLen;
  double sq_v = p.sfm_n_prime * B * theta, sq_a = p.sfm_n * B * theta;
  double fv = -std::exp(-norm(diff) / B - sq_v * sq_v);
  double fa = -(double)angle_sign(theta) * std::exp(-norm(diff) / B - sq_a * sq_a);
  return p.sfm_force_factor_social * (fv * interDir + fa * left_normal(interDir));
}
// lightsfm Group: member indices and their mean position.
struct Group {
  std::vector<size_t> members;
  V2 center;
};
// lightsfm computeGroupForce (non-_PAPER_VERSION_ branch; SURVEY.md Appendix A, UNPINNED):
//   gaze: if the centre of mass of the OTHER members is more than 90 deg off the
//         desired direction, pull along the desired direction;
//   coherence: relCOM * k_c * (tanh(|relCOM| - (n-1)/2) + 1) / 2;
//   repulsion: sum of (p_a - p_b) over overlapping members.
static V2 sfm_group(const sfw_params &p, size_t index, V2 desiredDir, const std::vector<Agent> &ag,
                    const std::map<int, Group> &groups) {
  const Agent &a = ag[index];
  auto it = groups.find(a.groupId);
  if (it == groups.end() || it->second.members.size() < 2) return {0, 0};
  const Group &g = it->second;
  const double n = (double)g.members.size();
  V2 gaze{0, 0};
  V2 com = (1.0 / (n - 1.0)) * (n * g.center - a.position);
  V2 rel = com - a.position;
  const double ep = desiredDir.x * rel.x + desiredDir.y * rel.y;
  const double comAngle = wrap_angle(std::acos(ep / (norm(desiredDir) * norm(rel))));
  if (comAngle > wrap_angle(90.0 * M_PI / 180.0)) {  // NaN (no desired direction) compares false
    const double dd2 = desiredDir.x * desiredDir.x + desiredDir.y * desiredDir.y;
    gaze = p.sfm_force_factor_group_gaze * ((ep / dd2) * desiredDir);
  }
  rel = g.center - a.position;
  const double dist = norm(rel), maxd = (n - 1.0) / 2.0;
  V2 coh = rel * (p.sfm_force_factor_group_coherence * (std::tanh(dist - maxd) + 1.0) / 2.0);
  V2 rep{0, 0};
  for (size_t m : g.members) {
    if (m == index) continue;
    V2 d = a.position - ag[m].position;
    if (norm(d) < a.radius + ag[m].radius) rep = rep + d;
  }
  rep = rep * p.sfm_force_factor_group_repulsion;
  return gaze + coh + rep;
}
// sfm::SFM.computeForces(std::vector<Agent>&)   (call site src/sfw_planner.cpp:592)
static void sfm_compute_forces(const sfw_params &p, std::vector<Agent> &ag) {
  std::map<int, Group> groups;
  for (size_t i = 0; i < ag.size(); ++i) {
    if (ag[i].groupId < 0) continue;
    Group &g = groups[ag[i].groupId];
    g.members.push_back(i);
    g.center = g.center + ag[i].position;
  }
  for (auto &kv : groups) kv.second.center = kv.second.center / (double)kv.second.members.size();
  for (size_t i = 0; i < ag.size(); ++i) {
    const V2 dir = sfm_desired(p, ag[i]);
    sfm_obstacle(p, ag[i]);
    ag[i].socialForce = {0, 0};
    for (size_t j = 0; j < ag.size(); ++j) {
      if (j == i) continue;
      ag[i].socialForce = ag[i].socialForce + sfm_pair(p, ag[i], ag[j]);
    }
    const V2 grp = groups.empty() ? V2{0, 0} : sfm_group(p, i, dir, ag, groups);
    ag[i].globalForce = ag[i].desiredForce + ag[i].socialForce + ag[i].obstacleForce + grp;
  }
}
// sfm::SFM.computeForces(Agent& me, std::vector<Agent>&)  (call site :697);
// skips by id, not by index.
static void sfm_compute_forces_single(const sfw_params &p, Agent &me, const std::vector<Agent> &others) {
  sfm_desired(p, me);
  sfm_obstacle(p, me);
  me.socialForce = {0, 0};
  for (const Agent &o : others) {
    if (o.id == me.id) continue;
    me.socialForce = me.socialForce + sfm_pair(p, me, o);
  }
  me.globalForce = me.desiredForce + me.socialForce + me.obstacleForce;
}
// sfm::SFM.updatePosition   (call site src/sfw_planner.cpp:594)
static void sfm_update_position(std::vector<Agent> &ag, double dt) {
  for (Agent &a : ag) {
    if (a.teleoperated) {
      double imd = a.linearVelocity * dt;
      double h = a.yaw + a.angularVelocity * dt * 0.5;
      a.position = a.position + V2{imd * std::cos(h), imd * std::sin(h)};
      a.yaw = wrap_angle(a.yaw + wrap_angle(a.angularVelocity * dt));
      a.velocity = {a.linearVelocity * std::cos(a.yaw), a.linearVelocity * std::sin(a.yaw)};
    } else {
      a.velocity = a.velocity + a.globalForce * dt;
      if (norm(a.velocity) > a.desiredVelocity)
        a.velocity = normalized(a.velocity) * a.desiredVelocity;
      a.yaw = angle_of(a.velocity);
      a.position = a.position + a.velocity * dt;
    }
    if (!a.goals.empty() && norm(a.goals.front().center - a.position) <= a.goals.front().radius) {
      Goal g = a.goals.front();
      a.goals.erase(a.goals.begin());
      if (a.cyclicGoals) a.goals.push_back(g);
    }
  }
}

// ---------------------------------------------------------------------------
// Planner helpers.  ref: include/.../sfw_planner.hpp:399-463
// ---------------------------------------------------------------------------
static inline float normalize_angle_f(float val, float mn, float mx) {
  if (val >= mn) return mn + std::fmod(val - mn, mx - mn);
  return mx - std::fmod(mn - val, mx - mn);
}
static inline double new_velocity(double vg, double vi, double a_max, double dt) {
  if ((vg - vi) >= 0) return std::min(vg, vi + a_max * dt);
  return std::max(vg, vi - a_max * dt);
}

// ref: src/sfw_planner.cpp:678-705
static double social_work(const sfw_params &p, const std::vector<Agent> &ag) {
  double wr = norm(ag[0].socialForce) + norm(ag[0].obstacleForce);
  std::vector<Agent> robot_only(1, ag[0]);
  double wp = 0.0;
  for (size_t i = 1; i < ag.size(); ++i) {
    Agent person = ag[i];
    sfm_compute_forces_single(p, person, robot_only);
    wp += norm(person.socialForce);
  }
  return wr + wp;
}

static inline int num_steps_of(const sfw_params &p) {
  int n = int(p.sim_time / p.sim_granularity + 0.5);  // ref :519
  return n == 0 ? 1 : n;                               // ref :523-525
}

// ref: src/sfw_planner.cpp:475-676.  pts (nullable) collects Trajectory points.
static double score_trajectory(const World &w, double x, double y, double theta, double vx,
                               double vy, double vtheta, double vx_samp, double vy_samp,
                               double vtheta_samp, double acc_x, double acc_y, double acc_theta,
                               double wpx, double wpy, std::vector<double> *pts) {
  const sfw_params &p = w.p;
  std::vector<Agent> my = w.agents;  // per-sample deep copy, ref :486
  double x_i = x, y_i = y, th_i = theta;
  double vx_i = vx, vy_i = vy, vth_i = vtheta;
  const int S = num_steps_of(p);
  const double dt = p.sim_time / S;  // ref :527
  double sw = 0.0, cm = 0.0;
  if (pts) pts->clear();
  const double rr = (double)(p.robot_radius * p.robot_radius);  // float product, ref :617

  for (int i = 0; i < S; ++i) {
    unsigned cx, cy;
    if (!w.worldToMap(x_i, y_i, cx, cy)) return -1.0;          // ref :545-550
    double fc = footprint_cost(w, x_i, y_i, th_i);             // ref :553
    if (fc >= 254.0) return -1.0;                              // ref :555
    if (fc < 0) return -1.0;                                   // ref :565
    cm += fc / 255.0;                                          // ref :575
    if (pts) { pts->push_back(x_i); pts->push_back(y_i); pts->push_back(th_i); }

    vx_i = new_velocity(vx_samp, vx_i, acc_x, dt);             // ref :581-583
    vy_i = new_velocity(vy_samp, vy_i, acc_y, dt);
    vth_i = new_velocity(vtheta_samp, vth_i, acc_theta, dt);
    // ref :586-588 — new velocities, OLD theta for both x and y
    const double xn = x_i + (vx_i * std::cos(th_i) + vy_i * std::cos(M_PI_2 + th_i)) * dt;
    const double yn = y_i + (vx_i * std::sin(th_i) + vy_i * std::sin(M_PI_2 + th_i)) * dt;
    x_i = xn; y_i = yn;
    th_i = th_i + vth_i * dt;

    if (!my.empty()) {
      sfm_compute_forces(p, my);                               // ref :592
      sfm_update_position(my, dt);                             // ref :594
      Agent &r = my[0];                                        // ref :600-610
      r.position = {x_i, y_i};
      r.yaw = wrap_angle(th_i);
      r.linearVelocity = hypotf((float)vx_i, (float)vy_i);
      r.angularVelocity = vth_i;
      r.velocity = {vx_i, vy_i};
      r.goals.assign(1, Goal{{wpx, wpy}, p.robot_goal_radius});
      for (size_t j = 1; j < my.size(); ++j) {                 // ref :613-627
        double dx = r.position.x - my[j].position.x, dy = r.position.y - my[j].position.y;
        if (dx * dx + dy * dy <= rr) return -1.0;
      }
      sw += social_work(p, my);                                // ref :629
    }
  }
  // ref :643-667
  const double dx = wpx - x_i, dy = wpy - y_i;
  const double d = dx * dx + dy * dy;
  double ang = std::atan2(dy, dx) - th_i;
  ang = normalize_angle_f((float)ang, (float)-M_PI, (float)M_PI);
  ang = std::fabs((float)ang) / M_PI;
  const double vel = std::fabs(p.max_vel_x - vx_i) / p.max_vel_x;
  cm = cm / S;
  return p.vel_weight * vel + p.distance_weight * d + p.angle_weight * ang +
         p.costmap_weight * cm + p.social_weight * sw;
}

// Sequential selection, literally as the reference scans.  ref :338-417, :426-468
static void select_best(const double *lin, int nv, const double *ang, int nw,
                        const double *costs, sfw_best *out) {
  double best_cost = 10000.0, bxv = 0.0, bth = 0.0, btc = -1.0;
  int64_t best_i = -1, n_valid = 0;
  double vx = 0.0, vt = 0.0;
  for (int iv = 0; iv < nv; ++iv)
    for (int iw = 0; iw < nw; ++iw) {
      const int64_t i = (int64_t)iv * nw + iw;
      const double linvel = lin[iv], angvel = ang[iw];
      if (linvel == 0.0 && angvel == 0.0) continue;
      const double c = costs[i];
      if (c >= 0.0) ++n_valid;
      if (c >= 0.0 && c <= best_cost) {
        if (c == best_cost && linvel < bxv) continue;
        if (c == best_cost && linvel == bxv && std::fabs(angvel) > std::fabs(bth)) continue;
        bxv = linvel; bth = angvel; btc = c;
        best_cost = c; best_i = i; vx = linvel; vt = angvel;
      }
    }
  out->n_valid = n_valid;
  if (btc != -1.0) { out->index = best_i; out->cost = btc; out->vx = vx; out->vy = 0.0; out->vtheta = vt; }
  else { out->index = -1; out->cost = -1.0; out->vx = out->vy = out->vtheta = 0.0; }
}


// ---------------------------------------------------------------------------
// findBestAction state machine around the grid loop.
// ref: src/sfw_planner.cpp:117-334, :426-468 (findBestAction), :853-892
// (updatePlan), :894-902 (isGoalReached / resetGoal).
// ---------------------------------------------------------------------------
struct CtrlParams {  // numeric ControllerParams, ref sfw_planner.hpp:186-226
  double max_vel_x, min_vel_x, max_vel_th, min_vel_th, max_trans_acc, max_rot_acc, min_in_place_vel_th;
  double yaw_goal_tolerance, xy_goal_tolerance, wp_tolerance;
  double sim_time, sim_granularity;
  double robot_radius;
  double social_weight, costmap_weight, angle_weight, distance_weight, vel_weight;
  int32_t is_circular, precision;
};
struct PlanPose { double x, y, yaw; };
enum Branch { kNotRunning = 0, kGoalReached, kRotateInPlace, kRotateBlocked, kApproach, kGrid, kGridFailed };

struct Planner {
  World *w = nullptr;
  CtrlParams c{};
  std::vector<double> lin, ang;
  std::vector<PlanPose> plan;
  int wp_index = -1;
  bool running = false, new_plan = false, goal_reached = false;
  double goal_x = 0, goal_y = 0, goal_t = 0;
  std::vector<double> last_costs;
  int last_branch = kNotRunning;

  void sync_params() {
    w->p.max_vel_x = c.max_vel_x;
    w->p.sim_time = c.sim_time;
    w->p.sim_granularity = c.sim_granularity;
    w->p.robot_radius = (float)c.robot_radius;
    w->p.social_weight = c.social_weight;
    w->p.costmap_weight = c.costmap_weight;
    w->p.angle_weight = c.angle_weight;
    w->p.distance_weight = c.distance_weight;
    w->p.vel_weight = c.vel_weight;
  }

  void update_plan(const PlanPose *poses, int n) {  // ref :853-892
    goal_reached = false;
    plan.assign(poses, poses + n);
    if (plan.empty()) { running = false; wp_index = -1; return; }
    wp_index = 0; running = true; new_plan = true;
    goal_x = plan.back().x; goal_y = plan.back().y; goal_t = plan.back().yaw;
  }

  // returns findBestAction's bool; cmd = (vx, vy, vtheta)
  bool find_best_action(const double pose[3], const double vel[3], double cmd[3]) {
    sync_params();
    goal_reached = false;
    double vx, vy = 0.0, vt;
    if (!running) { last_branch = kNotRunning; cmd[0] = cmd[1] = cmd[2] = 0.0; return true; }  // ref :131-142
    const float rx = (float)pose[0], ry = (float)pose[1], rt = (float)pose[2];             // ref :145-152
    const float rvx = (float)vel[0], rvy = (float)vel[1], rvt = (float)vel[2];
    const double dist_goal_sq = (rx - goal_x) * (rx - goal_x) + (ry - goal_y) * (ry - goal_y);
    if (dist_goal_sq < c.xy_goal_tolerance * c.xy_goal_tolerance) {                         // ref :176-233
      vx = 0.0;
      if (std::fabs(goal_t - rt) < c.yaw_goal_tolerance) {
        vt = 0.0; running = false; goal_reached = true; last_branch = kGoalReached;
      } else {
        float ad = (float)(goal_t - rt);
        ad = normalize_angle_f(ad, (float)-M_PI, (float)M_PI);
        vt = ad > 0.0f ? c.min_in_place_vel_th : -c.min_in_place_vel_th;
        last_branch = kRotateInPlace;
        if (!c.is_circular) {
          double sc = score_trajectory(*w, rx, ry, rt, rvx, rvy, rvt, vx, vy, vt, c.max_trans_acc, 0.0,
                                       c.max_rot_acc, 0.0, 0.0, nullptr);
          if (sc < 0.0) { last_branch = kRotateBlocked; cmd[0] = vx; cmd[1] = vy; cmd[2] = vt; return false; }
        }
      }
      cmd[0] = vx; cmd[1] = vy; cmd[2] = vt;
      return true;
    }
    if (new_plan) {                                                                          // ref :236-255
      new_plan = false;
      double min_dist = 9999.0;
      wp_index = 0;
      for (int i = (int)plan.size() - 1; i >= 0; --i) {
        const double dsq = (rx - plan[i].x) * (rx - plan[i].x) + (ry - plan[i].y) * (ry - plan[i].y);
        if (dsq < c.wp_tolerance * c.wp_tolerance) { wp_index = i; break; }
        else if (dsq < min_dist) { min_dist = dsq; wp_index = i; }
      }
    }
    double wpx = plan[wp_index].x, wpy = plan[wp_index].y;                                   // ref :258-271
    double dsw = (rx - wpx) * (rx - wpx) + (ry - wpy) * (ry - wpy);
    while (dsw < c.wp_tolerance * c.wp_tolerance && wp_index < (int)plan.size() - 1) {
      ++wp_index;
      wpx = plan[wp_index].x; wpy = plan[wp_index].y;
      dsw = (rx - wpx) * (rx - wpx) + (ry - wpy) * (ry - wpy);
    }
    const double dx = (wpx - rx) * std::cos(rt) + (wpy - ry) * std::sin(rt);                 // ref :274-276
    const double dy = -(wpx - rx) * std::sin(rt) + (wpy - ry) * std::cos(rt);
    const double dth = std::atan2(dy, dx);
    const double dist_thres = 1.5;                                                           // ref :282-334
    if (dist_goal_sq < dist_thres * dist_thres) {
      vx = c.min_vel_x + (c.max_vel_x - c.min_vel_x) * (std::sqrt(dist_goal_sq) / dist_thres);
      vy = 0.0;
      vt = c.min_vel_th + (c.max_vel_th - c.min_vel_th) * std::fabs(dth) / M_PI;
      if (dth < 0.0) vt *= -1;
      double sc = score_trajectory(*w, rx, ry, rt, rvx, rvy, rvt, vx, vy, vt, c.max_trans_acc, 0.0,
                                   c.max_rot_acc, wpx, wpy, nullptr);
      if (sc != -1) { last_branch = kApproach; cmd[0] = vx; cmd[1] = vy; cmd[2] = vt; return true; }
    }
    // the grid loop, ref :338-417
    const int nv = (int)lin.size(), nw = (int)ang.size();
    last_costs.assign((size_t)nv * nw, SFW_COST_INVALID);
    for (int iv = 0; iv < nv; ++iv)
      for (int iw = 0; iw < nw; ++iw) {
        const size_t i = (size_t)iv * nw + iw;
        if (lin[iv] == 0.0 && ang[iw] == 0.0) { last_costs[i] = SFW_COST_SKIPPED; continue; }
        last_costs[i] = score_trajectory(*w, rx, ry, rt, rvx, rvy, rvt, lin[iv], 0.0, ang[iw],
                                         c.max_trans_acc, 0.0, c.max_rot_acc, wpx, wpy, nullptr);
      }
    sfw_best b;
    select_best(lin.data(), nv, ang.data(), nw, last_costs.data(), &b);
    if (b.index >= 0) { last_branch = kGrid; cmd[0] = b.vx; cmd[1] = 0.0; cmd[2] = b.vtheta; return true; }
    last_branch = kGridFailed;                                                               // ref :456-468
    cmd[0] = cmd[1] = cmd[2] = 0.0;
    return false;
  }
};

}  // namespace sfwo

// ===========================================================================
// C entry points (same shapes as include/sfw_hip.h, prefix sfwo_)
// ===========================================================================
using sfwo::World;
extern "C" {

int sfwo_create(const sfw_params *p, void **out) {
  if (!p || !out) return SFW_ERR_INVALID_ARG;
  World *w = new World();
  w->p = *p;
  *out = w;
  return SFW_OK;
}
int sfwo_destroy(void *h) { delete static_cast<World *>(h); return SFW_OK; }
int sfwo_set_params(void *h, const sfw_params *p) {
  if (!h || !p) return SFW_ERR_INVALID_ARG;
  static_cast<World *>(h)->p = *p;
  return SFW_OK;
}
int sfwo_set_costmap(void *h, const uint8_t *cells, uint32_t sx, uint32_t sy, double ox,
                     double oy, double res) {
  if (!h || !cells || sx == 0 || sy == 0 || !(res > 0)) return SFW_ERR_INVALID_ARG;
  World *w = static_cast<World *>(h);
  w->cells.assign(cells, cells + (size_t)sx * sy);
  w->size_x = sx; w->size_y = sy; w->origin_x = ox; w->origin_y = oy; w->resolution = res;
  return SFW_OK;
}
int sfwo_set_footprint(void *h, const double *xy, int32_t K) {
  if (!h || K < 0 || (K > 0 && !xy)) return SFW_ERR_INVALID_ARG;
  World *w = static_cast<World *>(h);
  w->footprint.clear();
  for (int i = 0; i < K; ++i) w->footprint.push_back({xy[2 * i], xy[2 * i + 1]});
  return SFW_OK;
}
int sfwo_set_agents(void *h, const sfw_agent *a, int32_t A, const double *obs, int32_t O) {
  if (!h || A < 0 || O < 0 || (A > 0 && !a) || (O > 0 && !obs)) return SFW_ERR_INVALID_ARG;
  World *w = static_cast<World *>(h);
  w->obstacles.clear();
  for (int i = 0; i < O; ++i) w->obstacles.push_back({obs[2 * i], obs[2 * i + 1]});
  w->agents.clear();
  for (int i = 0; i < A; ++i) {
    sfwo::Agent g;
    g.position = {a[i].x, a[i].y};
    g.velocity = {a[i].vx, a[i].vy};
    g.desiredVelocity = a[i].desired_velocity;
    g.radius = a[i].radius;
    if (a[i].has_goal) g.goals.push_back({{a[i].goal_x, a[i].goal_y}, a[i].goal_radius});
    g.id = a[i].id;
    g.groupId = a[i].group_id;
    g.teleoperated = (i == 0);  // ref: src/sensor_interface.cpp:36, :489
    g.linearVelocity = sfwo::norm(g.velocity);
    g.obstacles1 = &w->obstacles;
    w->agents.push_back(g);
  }
  return SFW_OK;
}
static int check_world(const World *w) {
  if (w->cells.empty()) return SFW_ERR_STATE;
  return SFW_OK;
}
int sfwo_score_one(void *h, const sfw_robot_state *rs, double vxs, double vys, double vths,
                   const sfw_goal_args *g, double *cost_out, double *pts, int32_t cap,
                   int32_t *n_pts) {
  if (!h || !rs || !g || !cost_out) return SFW_ERR_INVALID_ARG;
  World *w = static_cast<World *>(h);
  if (int e = check_world(w)) return e;
  std::vector<double> tmp;
  *cost_out = sfwo::score_trajectory(*w, rs->x, rs->y, rs->theta, rs->vx, rs->vy, rs->vtheta, vxs,
                                     vys, vths, g->acc_x, g->acc_y, g->acc_theta, g->wpx, g->wpy,
                                     &tmp);
  int n = (int)(tmp.size() / 3);
  if (n_pts) *n_pts = n;
  if (pts) std::memcpy(pts, tmp.data(), sizeof(double) * 3 * std::min(n, (int)cap));
  return SFW_OK;
}
// n_threads <= 1: the reference's serial double loop.  > 1: one sample per
// OpenMP task (CPU baseline "all cores" variant).
int sfwo_score_grid(void *h, const sfw_robot_state *rs, const double *lin, int32_t nv,
                    const double *ang, int32_t nw, const sfw_goal_args *g, double *costs_out,
                    sfw_best *best_out, int32_t n_threads) {
  if (!h || !rs || !g || !lin || !ang || nv <= 0 || nw <= 0) return SFW_ERR_INVALID_ARG;
  World *w = static_cast<World *>(h);
  if (int e = check_world(w)) return e;
  const int64_t T = (int64_t)nv * nw;
  std::vector<double> local;
  double *costs = costs_out;
  if (!costs) { local.resize(T); costs = local.data(); }
#ifdef _OPENMP
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
#endif
  for (int64_t i = 0; i < T; ++i) {
    const double linvel = lin[i / nw], angvel = ang[i % nw];
    if (linvel == 0.0 && angvel == 0.0) { costs[i] = SFW_COST_SKIPPED; continue; }
    costs[i] = sfwo::score_trajectory(*w, rs->x, rs->y, rs->theta, rs->vx, rs->vy, rs->vtheta,
                                      linvel, 0.0, angvel, g->acc_x, g->acc_y, g->acc_theta,
                                      g->wpx, g->wpy, nullptr);
  }
  if (best_out) sfwo::select_best(lin, nv, ang, nw, costs, best_out);
  return SFW_OK;
}
// Exposed pieces for known-answer tests.
int sfwo_select_best(const double *lin, int32_t nv, const double *ang, int32_t nw,
                     const double *costs, sfw_best *out) {
  if (!lin || !ang || !costs || !out) return SFW_ERR_INVALID_ARG;
  sfwo::select_best(lin, nv, ang, nw, costs, out);
  return SFW_OK;
}
int sfwo_line_cells(int x0, int y0, int x1, int y1, int32_t *xy_out, int32_t cap) {
  int n = 0;
  sfwo::bresenham(x0, y0, x1, y1, [&](int x, int y) {
    if (n < cap) { xy_out[2 * n] = x; xy_out[2 * n + 1] = y; }
    ++n;
    return true;
  });
  return n;
}
double sfwo_footprint_cost(void *h, double x, double y, double th) {
  return sfwo::footprint_cost(*static_cast<World *>(h), x, y, th);
}
// Force on agent `me` from agent `other` (one social-force term).
int sfwo_pair_force(const sfw_params *p, const sfw_agent *me, const sfw_agent *other, double *fxy) {
  sfwo::Agent a, b;
  a.position = {me->x, me->y}; a.velocity = {me->vx, me->vy};
  b.position = {other->x, other->y}; b.velocity = {other->vx, other->vy};
  sfwo::V2 f = sfwo::sfm_pair(*p, a, b);
  fxy[0] = f.x; fxy[1] = f.y;
  return SFW_OK;
}
// Group force on agent `index` of a small agent set (positions/goals/groups), for KATs.
int sfwo_group_force(const sfw_params *p, const sfw_agent *a, int32_t A, int32_t index, double *fxy) {
  std::vector<sfwo::Agent> ag((size_t)A);
  std::map<int, sfwo::Group> groups;
  for (int i = 0; i < A; ++i) {
    ag[(size_t)i].position = {a[i].x, a[i].y};
    ag[(size_t)i].velocity = {a[i].vx, a[i].vy};
    ag[(size_t)i].radius = a[i].radius;
    ag[(size_t)i].desiredVelocity = a[i].desired_velocity;
    ag[(size_t)i].groupId = a[i].group_id;
    if (a[i].has_goal) ag[(size_t)i].goals.push_back({{a[i].goal_x, a[i].goal_y}, a[i].goal_radius});
    if (a[i].group_id >= 0) {
      groups[a[i].group_id].members.push_back((size_t)i);
      groups[a[i].group_id].center = groups[a[i].group_id].center + ag[(size_t)i].position;
    }
  }
  for (auto &kv : groups) kv.second.center = kv.second.center / (double)kv.second.members.size();
  const sfwo::V2 dir = sfwo::sfm_desired(*p, ag[(size_t)index]);
  const sfwo::V2 f = sfwo::sfm_group(*p, (size_t)index, dir, ag, groups);
  fxy[0] = f.x; fxy[1] = f.y;
  return SFW_OK;
}
float sfwo_normalize_angle(float v, float mn, float mx) { return sfwo::normalize_angle_f(v, mn, mx); }
int sfwo_num_steps(const sfw_params *p) { return sfwo::num_steps_of(*p); }
int sfwo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// ---- state machine ---------------------------------
void *sfwo_planner_create(void *world, const sfwo::CtrlParams *c) {
  sfwo::Planner *pl = new sfwo::Planner();
  pl->w = static_cast<World *>(world);
  pl->c = *c;
  // sample sets built once from the creation-time limits, ref :64-85
  const int n_lin = 4, n_ang = 4;
  const double ls = c->max_vel_x / n_lin, as = c->max_vel_th / n_ang;
  for (int i = 0; i <= n_lin; ++i) pl->lin.push_back(i * ls);
  pl->ang.push_back(0.0);
  for (int i = 1; i <= n_ang; ++i) { pl->ang.push_back(i * as); pl->ang.push_back(i * (-as)); }
  return pl;
}
void sfwo_planner_destroy(void *p) { delete static_cast<sfwo::Planner *>(p); }
void sfwo_planner_set_params(void *p, const sfwo::CtrlParams *c) { static_cast<sfwo::Planner *>(p)->c = *c; }
void sfwo_planner_set_sample_sets(void *p, const double *lin, int32_t nv, const double *ang, int32_t nw) {
  sfwo::Planner *pl = static_cast<sfwo::Planner *>(p);
  pl->lin.assign(lin, lin + nv);
  pl->ang.assign(ang, ang + nw);
}
void sfwo_planner_update_plan(void *p, const double *xyyaw, int32_t n) {
  std::vector<sfwo::PlanPose> v((size_t)n);
  for (int i = 0; i < n; ++i) v[i] = {xyyaw[3 * i], xyyaw[3 * i + 1], xyyaw[3 * i + 2]};
  static_cast<sfwo::Planner *>(p)->update_plan(v.data(), n);
}
int sfwo_planner_find_best_action(void *p, const double *pose, const double *vel, double *cmd,
                                  int32_t *found, int32_t *branch) {
  sfwo::Planner *pl = static_cast<sfwo::Planner *>(p);
  *found = pl->find_best_action(pose, vel, cmd) ? 1 : 0;
  *branch = pl->last_branch;
  return SFW_OK;
}
int sfwo_planner_is_goal_reached(void *p) {  // ref :894-900, one-shot
  sfwo::Planner *pl = static_cast<sfwo::Planner *>(p);
  if (pl->goal_reached) { pl->goal_reached = false; return 1; }
  return 0;
}
int sfwo_planner_wp_index(void *p) { return static_cast<sfwo::Planner *>(p)->wp_index; }
int sfwo_planner_running(void *p) { return static_cast<sfwo::Planner *>(p)->running ? 1 : 0; }
int64_t sfwo_planner_last_costs(void *p, double *out, int64_t cap) {
  const std::vector<double> &c = static_cast<sfwo::Planner *>(p)->last_costs;
  const int64_t n = (int64_t)c.size();
  if (out) std::memcpy(out, c.data(), sizeof(double) * (size_t)std::min(n, cap));
  return n;
}

// ---- sensor-interface restatement ------------------
// ref: src/sensor_interface.cpp laserCb :103-294, peopleCb :418-528, odomCb
// :534-581, getAgents :618-631.  One fixed 2-D transform stands in for tf.
struct sfwo_si {
  float max_robot_vel_x, robot_radius, person_radius, max_obstacle_dist, naive_goal_time, people_velocity;
  double tx, ty, yaw;
  bool tf_ok, running = false, odom_received = false;
  std::vector<sfw_agent> agents;
  std::vector<double> obstacles, agent_obstacles;
  std::vector<double> people_xy;  // last people message, controller frame unknown -> raw + frame flag
  bool people_in_ctrl = true;
};
void *sfwo_si_create(const float *p, double tx, double ty, double yaw, int32_t tf_ok) {
  sfwo_si *h = new sfwo_si();
  h->max_robot_vel_x = p[0]; h->robot_radius = p[1]; h->person_radius = p[2];
  h->max_obstacle_dist = p[3]; h->naive_goal_time = p[4]; h->people_velocity = p[5];
  h->tx = tx; h->ty = ty; h->yaw = yaw; h->tf_ok = tf_ok != 0;
  sfw_agent r;
  std::memset(&r, 0, sizeof(r));
  r.desired_velocity = h->max_robot_vel_x;  // ref :33-37
  r.radius = h->robot_radius;
  r.group_id = -1;
  r.id = SFW_ROBOT_ID_NONE;  // never set by the reference; a value no tracker tag takes (include/sfw_hip.h)
  h->agents.push_back(r);
  return h;
}
void sfwo_si_destroy(void *hv) { delete static_cast<sfwo_si *>(hv); }
void sfwo_si_start(void *hv) { static_cast<sfwo_si *>(hv)->running = true; }
void sfwo_si_stop(void *hv) { static_cast<sfwo_si *>(hv)->running = false; }
void sfwo_si_odom(void *hv, double x, double y, double yaw, double vx, double vy, double wz) {
  sfwo_si *h = static_cast<sfwo_si *>(hv);
  (void)yaw; (void)wz;
  if (!h->running) return;
  h->odom_received = true;
  h->agents[0].x = x; h->agents[0].y = y;
  h->agents[0].vx = vx; h->agents[0].vy = vy;  // local-frame twist, ref :565-575
}
void sfwo_si_people(void *hv, int32_t in_ctrl, const double *rows, const int32_t *ids, const int32_t *groups,
                    int32_t n) {
  sfwo_si *h = static_cast<sfwo_si *>(hv);
  if (!h->running || !h->odom_received) return;
  h->people_xy.clear();
  for (int i = 0; i < n; ++i) { h->people_xy.push_back(rows[6 * i]); h->people_xy.push_back(rows[6 * i + 1]); }
  h->people_in_ctrl = in_ctrl != 0;
  if (!in_ctrl && !h->tf_ok) return;
  const double c = std::cos(h->yaw), s = std::sin(h->yaw);
  std::vector<sfw_agent> out;
  for (int i = 0; i < n; ++i) {
    const double *r = rows + 6 * i;
    sfw_agent a;
    std::memset(&a, 0, sizeof(a));
    a.id = ids[i]; a.group_id = groups[i];
    if (in_ctrl) { a.x = r[0]; a.y = r[1]; a.vx = r[3]; a.vy = r[4]; }
    else {
      a.x = h->tx + c * r[0] - s * r[1]; a.y = h->ty + s * r[0] + c * r[1];
      a.vx = c * r[3] - s * r[4]; a.vy = s * r[3] + c * r[4];
    }
    a.radius = h->person_radius;
    a.goal_x = a.x + h->naive_goal_time * a.vx;   // ref :498-501
    a.goal_y = a.y + h->naive_goal_time * a.vy;
    a.goal_radius = h->person_radius;
    a.has_goal = 1;
    a.desired_velocity = h->people_velocity;
    out.push_back(a);
  }
  h->agents.resize((size_t)n + 1);
  h->agent_obstacles = h->obstacles;              // ref :513-524
  for (int i = 0; i < n; ++i) h->agents[(size_t)i + 1] = out[(size_t)i];
}
void sfwo_si_laser(void *hv, int32_t in_ctrl, float angle_min, float angle_inc, const float *ranges, int32_t n) {
  sfwo_si *h = static_cast<sfwo_si *>(hv);
  if (!h->running || !h->odom_received) return;
  std::vector<double> pts;
  float ang = angle_min;
  for (int i = 0; i < n; ++i) {
    const float r = ranges[i];
    if (!std::isnan(r) && std::isfinite(r) && r < h->max_obstacle_dist) {
      pts.push_back(r * std::cos(ang));  // float product, ref :125-126
      pts.push_back(r * std::sin(ang));
    }
    ang += angle_inc;
  }
  if (pts.empty()) { h->obstacles.clear(); return; }
  if (!in_ctrl && h->tf_ok) {
    const double c = std::cos(h->yaw), s = std::sin(h->yaw);
    for (size_t i = 0; i + 1 < pts.size(); i += 2) {
      const double x = pts[i], y = pts[i + 1];
      pts[i] = h->tx + c * x - s * y;
      pts[i + 1] = h->ty + s * x + c * y;
    }
  }
  std::vector<double> ppl = h->people_xy;
  if (!ppl.empty() && !h->people_in_ctrl) {
    if (!h->tf_ok) return;  // ref :195-202
    const double c = std::cos(h->yaw), s = std::sin(h->yaw);
    for (size_t i = 0; i + 1 < ppl.size(); i += 2) {
      const double x = ppl[i], y = ppl[i + 1];
      ppl[i] = h->tx + c * x - s * y;
      ppl[i + 1] = h->ty + s * x + c * y;
    }
  }
  if (!ppl.empty()) {  // ref :211-229
    std::vector<double> kept;
    for (size_t i = 0; i + 1 < pts.size(); i += 2) {
      bool rm = false;
      for (size_t j = 0; j + 1 < ppl.size() && !rm; j += 2) {
        const float dx = (float)(pts[i] - ppl[j]), dy = (float)(pts[i + 1] - ppl[j + 1]);
        rm = hypotf(dx, dy) <= h->person_radius;
      }
      if (!rm) { kept.push_back(pts[i]); kept.push_back(pts[i + 1]); }
    }
    pts.swap(kept);
  }
  h->obstacles = pts;
}
int32_t sfwo_si_get_agents(void *hv, sfw_agent *out, int32_t cap, double *obs_out, int32_t obs_cap, int32_t *O_out,
                           double *laser_out, int32_t laser_cap, int32_t *L_out) {
  sfwo_si *h = static_cast<sfwo_si *>(hv);
  const int A = (int)h->agents.size();
  for (int i = 0; i < A && i < cap; ++i) out[i] = h->agents[(size_t)i];
  const int O = (int)(h->agent_obstacles.size() / 2);
  for (int i = 0; i < 2 * O && i < 2 * obs_cap; ++i) obs_out[i] = h->agent_obstacles[(size_t)i];
  *O_out = O;
  const int Ln = (int)(h->obstacles.size() / 2);
  for (int i = 0; i < 2 * Ln && i < 2 * laser_cap; ++i) laser_out[i] = h->obstacles[(size_t)i];
  *L_out = Ln;
  return A;
}
}  // extern "C"
