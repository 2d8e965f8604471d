// sfw_host_capi.cpp — flat C shim over the C++ host mirror so that the Python
// tests (ctypes) can drive SFWPlanner::findBestAction / updatePlan exactly as
// nav2's controller_server would.  Test plumbing; the product interface is the
// C++ class in sfw_planner.hpp.
#include <cmath>
#include <cstring>
#include <exception>
#include <memory>
#include <string>

#include "sfw_planner.hpp"

using namespace social_force_window_planner;

extern "C" {

// Numeric ControllerParams, field for field (reference sfw_planner.hpp:186-226).
typedef struct sfwh_params {
  double max_vel_x, min_vel_x, max_vel_th, min_vel_th, max_trans_acc, max_rot_acc, min_in_place_vel_th;
  double yaw_goal_tolerance, xy_goal_tolerance, wp_tolerance;
  double sim_time, sim_granularity;
  double robot_radius;  // stored as float, like the reference
  double social_weight, costmap_weight, angle_weight, distance_weight, vel_weight;
  int32_t is_circular, precision;
} sfwh_params;

}  // extern "C"

namespace {

class StaticAgents : public AgentSource {
 public:
  AgentSet set;
  AgentSet getAgents() override { return set; }
};

struct HostHandle {
  std::shared_ptr<StaticAgents> agents = std::make_shared<StaticAgents>();
  std::vector<uint8_t> cells;
  CostmapView view;
  std::unique_ptr<SFWPlanner> planner;
  std::string err;
};

ControllerParams from_c(const sfwh_params &c) {
  ControllerParams p;
  p.max_vel_x_ = c.max_vel_x; p.min_vel_x_ = c.min_vel_x;
  p.max_vel_th_ = c.max_vel_th; p.min_vel_th_ = c.min_vel_th;
  p.max_trans_acc_ = c.max_trans_acc; p.max_rot_acc_ = c.max_rot_acc;
  p.min_in_place_vel_th_ = c.min_in_place_vel_th;
  p.yaw_goal_tolerance_ = c.yaw_goal_tolerance; p.xy_goal_tolerance_ = c.xy_goal_tolerance;
  p.wp_tolerance_ = c.wp_tolerance;
  p.sim_time_ = c.sim_time; p.sim_granularity_ = c.sim_granularity;
  p.robot_radius_ = static_cast<float>(c.robot_radius);
  p.social_weight_ = c.social_weight; p.costmap_weight_ = c.costmap_weight;
  p.angle_weight_ = c.angle_weight; p.distance_weight_ = c.distance_weight; p.vel_weight_ = c.vel_weight;
  p.is_circular_ = c.is_circular != 0;
  p.precision_ = c.precision;
  return p;
}

template <class F> int guarded(HostHandle *h, F &&f) {
  try {
    return f();
  } catch (const std::exception &e) {
    h->err = e.what();
    return -100;
  }
}

}  // namespace

extern "C" {

void sfwh_params_default(sfwh_params *c) {
  ControllerParams p;
  c->max_vel_x = p.max_vel_x_; c->min_vel_x = p.min_vel_x_;
  c->max_vel_th = p.max_vel_th_; c->min_vel_th = p.min_vel_th_;
  c->max_trans_acc = p.max_trans_acc_; c->max_rot_acc = p.max_rot_acc_;
  c->min_in_place_vel_th = p.min_in_place_vel_th_;
  c->yaw_goal_tolerance = p.yaw_goal_tolerance_; c->xy_goal_tolerance = p.xy_goal_tolerance_;
  c->wp_tolerance = p.wp_tolerance_;
  c->sim_time = p.sim_time_; c->sim_granularity = p.sim_granularity_;
  c->robot_radius = p.robot_radius_;
  c->social_weight = p.social_weight_; c->costmap_weight = p.costmap_weight_;
  c->angle_weight = p.angle_weight_; c->distance_weight = p.distance_weight_; c->vel_weight = p.vel_weight_;
  c->is_circular = p.is_circular_ ? 1 : 0;
  c->precision = p.precision_;
}

void *sfwh_create(const sfwh_params *c, const uint8_t *cells, uint32_t sx, uint32_t sy, double ox, double oy,
                  double res, const double *footprint_xy, int32_t K, int32_t device) {
  HostHandle *h = new HostHandle();
  h->cells.assign(cells, cells + static_cast<size_t>(sx) * sy);
  h->view = CostmapView{h->cells.data(), sx, sy, ox, oy, res};
  std::vector<Point> fp;
  for (int i = 0; i < K; ++i) fp.push_back(Point{footprint_xy[2 * i], footprint_xy[2 * i + 1], 0.0});
  h->planner.reset(new SFWPlanner(from_c(*c), h->agents, h->view, fp, device));
  return h;
}
void sfwh_destroy(void *hv) { delete static_cast<HostHandle *>(hv); }
const char *sfwh_last_error(void *hv) { return static_cast<HostHandle *>(hv)->err.c_str(); }

int sfwh_set_params(void *hv, const sfwh_params *c) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  return guarded(h, [&] { h->planner->setParams(from_c(*c)); return 0; });
}
int sfwh_set_costmap(void *hv, const uint8_t *cells, uint32_t sx, uint32_t sy, double ox, double oy, double res) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  h->cells.assign(cells, cells + static_cast<size_t>(sx) * sy);
  h->view = CostmapView{h->cells.data(), sx, sy, ox, oy, res};
  h->planner->setCostmap(h->view);
  return 0;
}
int sfwh_set_agents(void *hv, const sfw_agent *agents, int32_t A, const double *obs_xy, int32_t O) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  h->agents->set.agents.assign(agents, agents + A);
  h->agents->set.obstacles_xy.assign(obs_xy, obs_xy + 2 * static_cast<size_t>(O));
  return 0;
}
// Multi-device mode (SFWPlanner::setDevices); before the first scoring call.
int sfwh_set_devices(void *hv, const int *devices, int32_t n, int32_t host_reduce) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  try {
    h->planner->setDevices(std::vector<int>(devices, devices + n), host_reduce != 0);
  } catch (const std::exception &e) {
    h->err = e.what();
    return -1;
  }
  return 0;
}
int sfwh_ranks(void *hv) { return static_cast<HostHandle *>(hv)->planner->ranks(); }
int sfwh_set_sample_sets(void *hv, const double *lin, int32_t nv, const double *ang, int32_t nw) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  h->planner->setSampleSets(std::vector<double>(lin, lin + nv), std::vector<double>(ang, ang + nw));
  return 0;
}
// plan: n poses as (x, y, yaw) triples
int sfwh_update_plan(void *hv, const double *xyyaw, int32_t n) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  std::vector<PoseStamped> plan(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    plan[i].pose.position.x = xyyaw[3 * i];
    plan[i].pose.position.y = xyyaw[3 * i + 1];
    plan[i].pose.orientation = quaternionFromYaw(xyyaw[3 * i + 2]);
  }
  return guarded(h, [&] { return h->planner->updatePlan(plan) ? 0 : 1; });
}
// pose = (x, y, yaw), vel = (vx, vy, vtheta); cmd_out = (vx, vy, vtheta).
// *found_out = return value of findBestAction; *branch_out = SFWPlanner::Branch.
int sfwh_find_best_action(void *hv, const double *pose, const double *vel, double *cmd_out,
                          int32_t *found_out, int32_t *branch_out) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  return guarded(h, [&] {
    PoseStamped ps;
    ps.pose.position.x = pose[0];
    ps.pose.position.y = pose[1];
    ps.pose.orientation = quaternionFromYaw(pose[2]);
    Twist tw, cmd;
    tw.linear.x = vel[0];
    tw.linear.y = vel[1];
    tw.angular.z = vel[2];
    const bool ok = h->planner->findBestAction(ps, tw, cmd);
    cmd_out[0] = cmd.linear.x;
    cmd_out[1] = cmd.linear.y;
    cmd_out[2] = cmd.angular.z;
    *found_out = ok ? 1 : 0;
    *branch_out = h->planner->lastBranch();
    return 0;
  });
}
int sfwh_is_goal_reached(void *hv) { return static_cast<HostHandle *>(hv)->planner->isGoalReached() ? 1 : 0; }
int sfwh_wp_index(void *hv) { return static_cast<HostHandle *>(hv)->planner->wpIndex(); }
int sfwh_running(void *hv) { return static_cast<HostHandle *>(hv)->planner->running() ? 1 : 0; }
int64_t sfwh_last_costs(void *hv, double *out, int64_t cap) {
  const std::vector<double> &c = static_cast<HostHandle *>(hv)->planner->lastCosts();
  const int64_t n = static_cast<int64_t>(c.size());
  if (out) std::memcpy(out, c.data(), sizeof(double) * static_cast<size_t>(n < cap ? n : cap));
  return n;
}
int sfwh_trajectory_points(void *hv, int64_t index, double *xyth, int32_t cap) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  return guarded(h, [&] {
    Trajectory t;
    if (!h->planner->getTrajectoryPoints(index, t)) return -1;
    const int n = static_cast<int>(t.getPointsSize());
    for (int i = 0; i < n && i < cap; ++i) t.getPoint(i, xyth[3 * i], xyth[3 * i + 1], xyth[3 * i + 2]);
    return n;
  });
}
// all trajectories of the last grid: xyth = T x cap x 3, counts = T; returns T or -1
int64_t sfwh_all_trajectories(void *hv, double *xyth, int32_t cap, int32_t *counts) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  std::vector<Trajectory> ts;
  if (!h->planner->getTrajectories(ts)) return -1;
  for (size_t i = 0; i < ts.size(); ++i) {
    const int n = static_cast<int>(ts[i].getPointsSize());
    counts[i] = n;
    for (int k = 0; k < n && k < cap; ++k)
      ts[i].getPoint(k, xyth[(i * cap + k) * 3], xyth[(i * cap + k) * 3 + 1], xyth[(i * cap + k) * 3 + 2]);
  }
  return static_cast<int64_t>(ts.size());
}
// markers of the last cycle: rgba = cap x 4 floats, counts = cap point counts, z0 = cap z of each marker's first point (0 if
// none); returns T, or -1 when the cycle left the markers untouched.  Nothing is written when cap < T.
int64_t sfwh_markers(void *hv, float *rgba, int32_t *counts, double *z0, int64_t cap) {
  HostHandle *h = static_cast<HostHandle *>(hv);
  std::vector<SFWPlanner::MarkerData> ms;
  if (!h->planner->getMarkers(ms)) return -1;
  if (static_cast<int64_t>(ms.size()) > cap) return static_cast<int64_t>(ms.size());
  for (size_t i = 0; i < ms.size(); ++i) {
    rgba[4 * i] = ms[i].r; rgba[4 * i + 1] = ms[i].g; rgba[4 * i + 2] = ms[i].b; rgba[4 * i + 3] = ms[i].a;
    counts[i] = static_cast<int32_t>(ms[i].points.size());
    z0[i] = ms[i].points.empty() ? 0.0 : ms[i].points[0].z;
  }
  return static_cast<int64_t>(ms.size());
}
void sfwh_set_marker_capture(void *hv, int32_t on) { static_cast<HostHandle *>(hv)->planner->setMarkerCapture(on != 0); }
double sfwh_get_yaw(double x, double y, double z, double w) { return getYaw(Quaternion{x, y, z, w}); }
}

// ---------------------------------------------------------------------------
// SFMSensorInterface shim.  One fixed transform stands in
// for tf: every frame other than "odom" maps to the controller frame through it.
// ---------------------------------------------------------------------------
#include "sensor_interface.hpp"

namespace {
struct SiHandle {
  Transform2D tf;
  bool tf_ok = true;
  std::unique_ptr<SFMSensorInterface> si;
};
}  // namespace

extern "C" {

// p = {max_robot_vel_x, robot_radius, person_radius, max_obstacle_dist, naive_goal_time, people_velocity}
void *sfwh_si_create(const float *p, double tx, double ty, double yaw, int32_t tf_ok) {
  SiHandle *h = new SiHandle();
  h->tf = Transform2D{tx, ty, yaw};
  h->tf_ok = tf_ok != 0;
  InterfaceParams ip;
  ip.max_robot_vel_x_ = p[0]; ip.robot_radius_ = p[1]; ip.person_radius_ = p[2];
  ip.max_obstacle_dist_ = p[3]; ip.naive_goal_time_ = p[4]; ip.people_velocity_ = p[5];
  SiHandle *raw = h;
  h->si.reset(new SFMSensorInterface(ip, [raw](const std::string &, const std::string &, Transform2D &out) {
    out = raw->tf;
    return raw->tf_ok;
  }));
  return h;
}
void sfwh_si_destroy(void *hv) { delete static_cast<SiHandle *>(hv); }
void sfwh_si_start(void *hv) { static_cast<SiHandle *>(hv)->si->start(); }
void sfwh_si_stop(void *hv) { static_cast<SiHandle *>(hv)->si->stop(); }
void sfwh_si_odom(void *hv, double x, double y, double yaw, double vx, double vy, double wz) {
  Odometry o;
  o.frame_id = "odom";
  o.pose.position.x = x;
  o.pose.position.y = y;
  o.pose.orientation = quaternionFromYaw(yaw);
  o.twist.linear.x = vx;
  o.twist.linear.y = vy;
  o.twist.angular.z = wz;
  static_cast<SiHandle *>(hv)->si->odomCb(o);
}
// rows: n x {x, y, yaw, vx, vy, wz}; ids/groups: n
void sfwh_si_people(void *hv, int32_t in_controller_frame, const double *rows, const int32_t *ids,
                    const int32_t *groups, int32_t n) {
  People pp;
  pp.frame_id = in_controller_frame ? "odom" : "map";
  for (int i = 0; i < n; ++i) {
    Person p;
    p.position.x = rows[6 * i];
    p.position.y = rows[6 * i + 1];
    p.position.z = rows[6 * i + 2];
    p.velocity.x = rows[6 * i + 3];
    p.velocity.y = rows[6 * i + 4];
    p.velocity.z = rows[6 * i + 5];
    p.tags = {std::to_string(ids[i]), std::to_string(groups[i])};
    pp.people.push_back(p);
  }
  static_cast<SiHandle *>(hv)->si->peopleCb(pp);
}
void sfwh_si_laser(void *hv, int32_t in_controller_frame, float angle_min, float angle_inc, const float *ranges,
                   int32_t n) {
  LaserScan s;
  s.frame_id = in_controller_frame ? "odom" : "base_laser";
  s.angle_min = angle_min;
  s.angle_increment = angle_inc;
  s.ranges.assign(ranges, ranges + n);
  static_cast<SiHandle *>(hv)->si->laserCb(s);
}
// returns A; *O_out = number of obstacle points the agents carry; *L_out = points of the last scan
int32_t sfwh_si_get_agents(void *hv, sfw_agent *out, int32_t cap, double *obs_out, int32_t obs_cap, int32_t *O_out,
                           double *laser_out, int32_t laser_cap, int32_t *L_out) {
  SiHandle *h = static_cast<SiHandle *>(hv);
  const AgentSet s = h->si->getAgents();
  const int A = static_cast<int>(s.agents.size());
  for (int i = 0; i < A && i < cap; ++i) out[i] = s.agents[i];
  const int O = static_cast<int>(s.obstacles_xy.size() / 2);
  for (int i = 0; i < 2 * O && i < 2 * obs_cap; ++i) obs_out[i] = s.obstacles_xy[i];
  *O_out = O;
  const std::vector<double> &l = h->si->obstacles();
  const int Ln = static_cast<int>(l.size() / 2);
  for (int i = 0; i < 2 * Ln && i < 2 * laser_cap; ++i) laser_out[i] = l[i];
  *L_out = Ln;
  return A;
}
}

// ---------------------------------------------------------------------------
// transformGlobalPlan shim: plan as (x,y,yaw) triples; the costmap-frame transform
// is a rigid 2-D transform (tx,ty,yaw).  Returns the number of poses written to
// out (capacity cap) and the new length of the stored plan in *remaining;
// -1 / -2 for the two PlannerException cases.
// ---------------------------------------------------------------------------
#include "plan_utils.hpp"

extern "C" int32_t sfwh_transform_global_plan(double *plan_xyyaw, int32_t n, const double *robot_xy, uint32_t sx,
                                              uint32_t sy, double resolution, double tx, double ty, double yaw,
                                              double *out_xyyaw, int32_t cap, int32_t *remaining) {
  std::vector<PoseStamped> plan(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) {
    plan[i].pose.position.x = plan_xyyaw[3 * i];
    plan[i].pose.position.y = plan_xyyaw[3 * i + 1];
    plan[i].pose.orientation = quaternionFromYaw(plan_xyyaw[3 * i + 2]);
  }
  PoseStamped robot;
  robot.pose.position.x = robot_xy[0];
  robot.pose.position.y = robot_xy[1];
  const double c = std::cos(yaw), s = std::sin(yaw);
  std::vector<PoseStamped> out;
  try {
    out = transformGlobalPlan(plan, robot, sx, sy, resolution, [&](const PoseStamped &in, PoseStamped &o) {
      o = in;
      o.pose.position.x = tx + c * in.pose.position.x - s * in.pose.position.y;
      o.pose.position.y = ty + s * in.pose.position.x + c * in.pose.position.y;
      o.pose.orientation = quaternionFromYaw(getYaw(in.pose.orientation) + yaw);
      return true;
    });
  } catch (const PlannerException &e) {
    *remaining = static_cast<int32_t>(plan.size());
    return std::string(e.what()).find("zero length") != std::string::npos ? -1 : -2;
  }
  *remaining = static_cast<int32_t>(plan.size());
  for (size_t i = 0; i < plan.size(); ++i) {
    plan_xyyaw[3 * i] = plan[i].pose.position.x;
    plan_xyyaw[3 * i + 1] = plan[i].pose.position.y;
    plan_xyyaw[3 * i + 2] = getYaw(plan[i].pose.orientation);
  }
  const int m = static_cast<int>(out.size());
  for (int i = 0; i < m && i < cap; ++i) {
    out_xyyaw[3 * i] = out[i].pose.position.x;
    out_xyyaw[3 * i + 1] = out[i].pose.position.y;
    out_xyyaw[3 * i + 2] = getYaw(out[i].pose.orientation);
  }
  return m;
}
