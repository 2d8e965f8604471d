// sfw_planner.cpp — host mirror of the reference planner core over the C ABI.
// Control flow follows reference src/sfw_planner.cpp:117-468 (findBestAction),
// :853-892 (updatePlan), :894-902 (isGoalReached/resetGoal); the scoring itself
// (reference :338-417 grid loop, :475-705 scoreTrajectory/computeSocialWork) is
// executed by the HIP kernels behind include/sfw_hip.h.
#include "sfw_planner.hpp"

#include <cmath>
#include <stdexcept>

namespace social_force_window_planner {

double getYaw(const Quaternion &q) {
  // yaw of a unit quaternion (tf2::getYaw, tf2/utils.h)
  const double sqx = q.x * q.x, sqy = q.y * q.y, sqz = q.z * q.z, sqw = q.w * q.w;
  const double sarg = -2.0 * (q.x * q.z - q.w * q.y) / (sqx + sqy + sqz + sqw);
  if (sarg <= -0.99999) return -2.0 * std::atan2(q.y, q.x);
  if (sarg >= 0.99999) return 2.0 * std::atan2(q.y, q.x);
  return std::atan2(2.0 * (q.x * q.y + q.w * q.z), sqw + sqx - sqy - sqz);
}
Quaternion quaternionFromYaw(double yaw) {
  Quaternion q;
  q.z = std::sin(yaw * 0.5);
  q.w = std::cos(yaw * 0.5);
  return q;
}

sfw_params ControllerParams::toAbi() const {
  sfw_params p;
  sfw_params_default(&p);  // lightsfm defaults: the reference never overrides them
  p.max_vel_x = max_vel_x_;
  p.sim_time = sim_time_;
  p.sim_granularity = sim_granularity_;
  p.robot_radius = robot_radius_;
  p.social_weight = social_weight_;
  p.costmap_weight = costmap_weight_;
  p.angle_weight = angle_weight_;
  p.distance_weight = distance_weight_;
  p.vel_weight = vel_weight_;
  p.precision = precision_;
  return p;
}

// ---- Trajectory -----------------------------------------------------------
Trajectory::Trajectory() : xv_(0.0), yv_(0.0), thetav_(0.0), cost_(-1.0) {}
Trajectory::Trajectory(double xv, double yv, double thetav, double time_delta, unsigned int num_pts)
    : xv_(xv), yv_(yv), thetav_(thetav), cost_(-1.0), time_delta_(time_delta), pts_(num_pts) {}
void Trajectory::getPoint(unsigned int i, double &x, double &y, double &th) const {
  x = pts_[i].x; y = pts_[i].y; th = pts_[i].th;
}
void Trajectory::setPoint(unsigned int i, double x, double y, double th) { pts_[i] = {x, y, th}; }
void Trajectory::addPoint(double x, double y, double th) { pts_.push_back({x, y, th}); }
void Trajectory::getEndpoint(double &x, double &y, double &th) const {
  x = pts_.back().x; y = pts_.back().y; th = pts_.back().th;
}
void Trajectory::resetPoints() { pts_.clear(); }
unsigned int Trajectory::getPointsSize() const { return static_cast<unsigned int>(pts_.size()); }

void referenceSampleSets(double max_vel_x, double max_vel_th, std::vector<double> &lin,
                         std::vector<double> &ang) {
  // reference src/sfw_planner.cpp:64-85
  const int n_lin = 4, n_ang = 4;
  const double ls = max_vel_x / n_lin, as = max_vel_th / n_ang;
  lin.clear();
  ang.clear();
  for (int i = 0; i <= n_lin; ++i) lin.push_back(i * ls);
  ang.push_back(0.0);
  for (int i = 1; i <= n_ang; ++i) {
    ang.push_back(i * as);
    ang.push_back(i * (-as));
  }
}

namespace {
// reference sfw_planner.hpp:399-407
inline float normalizeAngle(float val, float mn, float mx) {
  if (val >= mn) return mn + std::fmod(val - mn, mx - mn);
  return mx - std::fmod(mn - val, mx - mn);
}
inline void setCmd(Twist &c, double vx, double vy, double vt) {
  c.linear.x = vx; c.linear.y = vy; c.linear.z = 0.0;
  c.angular.x = 0.0; c.angular.y = 0.0; c.angular.z = vt;
}
}  // namespace

// ---- SFWPlanner -----------------------------------------------------------
SFWPlanner::SFWPlanner(const ControllerParams &params, std::shared_ptr<AgentSource> sensor_iface,
                       const CostmapView &costmap, std::vector<Point> footprint_spec, int device)
    : params_(params), sensor_iface_(std::move(sensor_iface)), costmap_(costmap),
      footprint_spec_(std::move(footprint_spec)), device_(device) {
  // sample sets are built once from the constructor-time limits (:64-85)
  referenceSampleSets(params_.max_vel_x_, params_.max_vel_th_, linvels_, angvels_);
}

// The device handle is created on the first scoring call, so the branches of
// findBestAction that never score (not running, goal reached, circular rotate
// in place) work on a host without a GPU; every scoring branch throws there.
void SFWPlanner::ensureDevice() {
  if (handle_) return;
  const sfw_params abi = params_.toAbi();
  if (!devices_.empty()) {
    const int rc = sfw_multi_create(&abi, devices_.data(), static_cast<int32_t>(devices_.size()),
                                    host_reduce_ ? SFW_MULTI_HOST_REDUCE : SFW_MULTI_RCCL, &multi_);
    if (rc != SFW_OK) {
      multi_ = nullptr;
      raise("sfw_multi_create (devices not visible, listed twice without host_reduce, or librccl.so missing)", rc);
    }
    handle_ = sfw_multi_rank_handle(multi_, 0);
    setMarkerCapture(marker_capture_);
    return;
  }
  const int rc = sfw_create(&abi, device_, &handle_);
  if (rc != SFW_OK) {
    handle_ = nullptr;
    raise("sfw_create (no HIP device? this planner has no CPU scoring path)", rc);
  }
  setMarkerCapture(marker_capture_);
}

void SFWPlanner::setDevices(std::vector<int> devices, bool host_reduce) {
  if (handle_) throw std::runtime_error("SFWPlanner::setDevices: the device handle already exists");
  devices_ = std::move(devices);
  host_reduce_ = host_reduce;
}

SFWPlanner::~SFWPlanner() {
  if (multi_) sfw_multi_destroy(multi_);
  else if (handle_) sfw_destroy(handle_);
}

void SFWPlanner::raise(const char *what, int status) const {
  std::string msg = std::string("SFWPlanner: ") + what + " failed with status " + std::to_string(status);
  if (multi_ && *sfw_multi_last_error(multi_)) msg += std::string(": ") + sfw_multi_last_error(multi_);
  else if (handle_) msg += std::string(": ") + sfw_last_error(handle_);
  throw std::runtime_error(msg);
}

void SFWPlanner::setParams(const ControllerParams &p) {
  std::lock_guard<std::mutex> l(configuration_mutex_);
  params_ = p;
}
void SFWPlanner::setMarkerCapture(bool on) {
  marker_capture_ = on;
  if (multi_) {
    for (int32_t r = 0; r < sfw_multi_ranks(multi_); ++r) sfw_set_points_capture(sfw_multi_rank_handle(multi_, r), on ? 1 : 0);
  } else if (handle_) {
    sfw_set_points_capture(handle_, on ? 1 : 0);
  }
}
void SFWPlanner::setFootprint(std::vector<Point> footprint) { footprint_spec_ = std::move(footprint); }
void SFWPlanner::setSampleSets(std::vector<double> lin, std::vector<double> ang) {
  linvels_ = std::move(lin);
  angvels_ = std::move(ang);
}

void SFWPlanner::uploadWorld(const AgentSet &agents) {
  ensureDevice();
  const sfw_params abi = params_.toAbi();
  std::vector<double> fp;
  fp.reserve(2 * footprint_spec_.size());
  for (const Point &q : footprint_spec_) { fp.push_back(q.x); fp.push_back(q.y); }
  const double *fpp = fp.empty() ? nullptr : fp.data();
  const int32_t K = static_cast<int32_t>(footprint_spec_.size());
  const sfw_agent *ag = agents.agents.empty() ? nullptr : agents.agents.data();
  const int32_t A = static_cast<int32_t>(agents.agents.size());
  const double *obs = agents.obstacles_xy.empty() ? nullptr : agents.obstacles_xy.data();
  const int32_t O = static_cast<int32_t>(agents.obstacles_xy.size() / 2);
  int rc;
  if (multi_) {  // replicated to every rank
    if ((rc = sfw_multi_set_params(multi_, &abi)) != SFW_OK) raise("sfw_multi_set_params", rc);
    if ((rc = sfw_multi_set_costmap(multi_, costmap_.cells, costmap_.size_x, costmap_.size_y, costmap_.origin_x,
                                    costmap_.origin_y, costmap_.resolution)) != SFW_OK)
      raise("sfw_multi_set_costmap", rc);
    if ((rc = sfw_multi_set_footprint(multi_, fpp, K)) != SFW_OK) raise("sfw_multi_set_footprint", rc);
    if ((rc = sfw_multi_set_agents(multi_, ag, A, obs, O)) != SFW_OK) raise("sfw_multi_set_agents", rc);
    return;
  }
  if ((rc = sfw_set_params(handle_, &abi)) != SFW_OK) raise("sfw_set_params", rc);
  if ((rc = sfw_set_costmap(handle_, costmap_.cells, costmap_.size_x, costmap_.size_y, costmap_.origin_x,
                            costmap_.origin_y, costmap_.resolution)) != SFW_OK)
    raise("sfw_set_costmap", rc);
  if ((rc = sfw_set_footprint(handle_, fpp, K)) != SFW_OK) raise("sfw_set_footprint", rc);
  if ((rc = sfw_set_agents(handle_, ag, A, obs, O)) != SFW_OK) raise("sfw_set_agents", rc);
}

double SFWPlanner::scoreTrajectory(double x, double y, double theta, double vx, double vy, double vtheta,
                                   double vx_samp, double vy_samp, double vtheta_samp, double acc_x,
                                   double acc_y, double acc_theta, double wpx, double wpy,
                                   const AgentSet &agents, Trajectory &traj) {
  uploadWorld(agents);
  const sfw_robot_state rs{x, y, theta, vx, vy, vtheta};
  const sfw_goal_args ga{acc_x, acc_y, acc_theta, wpx, wpy};
  int S = static_cast<int>(params_.sim_time_ / params_.sim_granularity_ + 0.5);
  if (S == 0) S = 1;
  std::vector<double> pts(static_cast<size_t>(3) * S);
  double cost = -1.0;
  int32_t n = 0;
  const int rc = sfw_score_one(handle_, &rs, vx_samp, vy_samp, vtheta_samp, &ga, &cost, pts.data(), S, &n);
  if (rc != SFW_OK) raise("sfw_score_one", rc);
  grid_staged_ = false;
  traj.resetPoints();                       // ref :531-535
  traj.xv_ = vx_samp;
  traj.yv_ = vy_samp;
  traj.thetav_ = vtheta_samp;
  for (int i = 0; i < n; ++i) traj.addPoint(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  traj.cost_ = cost;                        // -1.0 when invalid (ref :546,:556,:567,:624,:674)
  return cost;
}

bool SFWPlanner::findBestAction(const PoseStamped &global_pose, const Twist &global_vel, Twist &cmd_vel) {
  std::lock_guard<std::mutex> lock(configuration_mutex_);  // ref :123
  goal_reached_ = false;                                   // ref :127
  double vx, vy = 0.0, vt;

  if (!running_) {                                         // ref :131-142
    last_branch_ = kNotRunning;
    setCmd(cmd_vel, 0.0, 0.0, 0.0);
    return true;
  }

  // pose and velocity are truncated to float (ref :145-152)
  const float rx = static_cast<float>(global_pose.pose.position.x);
  const float ry = static_cast<float>(global_pose.pose.position.y);
  const float rt = static_cast<float>(getYaw(global_pose.pose.orientation));
  const float rvx = static_cast<float>(global_vel.linear.x);
  const float rvy = static_cast<float>(global_vel.linear.y);
  const float rvt = static_cast<float>(global_vel.angular.z);

  AgentSet agents;
  if (sensor_iface_) agents = sensor_iface_->getAgents();  // ref :156

  const double dist_goal_sq = (rx - goal_x_) * (rx - goal_x_) + (ry - goal_y_) * (ry - goal_y_);  // ref :167

  if (dist_goal_sq < params_.xy_goal_tolerance_ * params_.xy_goal_tolerance_) {  // ref :176-233
    vx = 0.0;
    if (std::fabs(goal_t_ - rt) < params_.yaw_goal_tolerance_) {
      vt = 0.0;
      running_ = false;
      goal_reached_ = true;
      last_branch_ = kGoalReached;
    } else {
      float ang_diff = static_cast<float>(goal_t_ - rt);
      ang_diff = normalizeAngle(ang_diff, static_cast<float>(-M_PI), static_cast<float>(M_PI));
      vt = (ang_diff > 0.0f) ? params_.min_in_place_vel_th_ : -params_.min_in_place_vel_th_;
      last_branch_ = kRotateInPlace;
      if (!params_.is_circular_) {
        Trajectory t;
        if (scoreTrajectory(rx, ry, rt, rvx, rvy, rvt, vx, vy, vt, params_.max_trans_acc_, 0.0,
                            params_.max_rot_acc_, 0.0, 0.0, agents, t) < 0.0) {
          last_branch_ = kRotateBlocked;
          setCmd(cmd_vel, vx, vy, vt);
          return false;
        }
      }
    }
    setCmd(cmd_vel, vx, vy, vt);
    return true;
  }

  if (new_plan_) {  // ref :236-255: nearest way-point, searched from the END of the plan
    new_plan_ = false;
    double min_dist = 9999.0;
    wp_index_ = 0;
    for (int i = static_cast<int>(global_plan_.size()) - 1; i >= 0; --i) {
      const double wx = global_plan_[i].pose.position.x, wy = global_plan_[i].pose.position.y;
      const double dsq = (rx - wx) * (rx - wx) + (ry - wy) * (ry - wy);
      if (dsq < params_.wp_tolerance_ * params_.wp_tolerance_) {
        wp_index_ = i;
        break;
      } else if (dsq < min_dist) {
        min_dist = dsq;
        wp_index_ = i;
      }
    }
  }

  double wpx = global_plan_[wp_index_].pose.position.x, wpy = global_plan_[wp_index_].pose.position.y;
  double dist_swp_sq = (rx - wpx) * (rx - wpx) + (ry - wpy) * (ry - wpy);
  while (dist_swp_sq < params_.wp_tolerance_ * params_.wp_tolerance_ &&
         wp_index_ < static_cast<int>(global_plan_.size()) - 1) {  // ref :264-271
    ++wp_index_;
    wpx = global_plan_[wp_index_].pose.position.x;
    wpy = global_plan_[wp_index_].pose.position.y;
    dist_swp_sq = (rx - wpx) * (rx - wpx) + (ry - wpy) * (ry - wpy);
  }

  // way-point in the robot frame (ref :274-276)
  const double dx = (wpx - rx) * std::cos(rt) + (wpy - ry) * std::sin(rt);
  const double dy = -(wpx - rx) * std::sin(rt) + (wpy - ry) * std::cos(rt);
  const double dth = std::atan2(dy, dx);

  const double dist_thres = 1.5;  // ref :282-334
  if (dist_goal_sq < dist_thres * dist_thres) {
    vx = params_.min_vel_x_ + (params_.max_vel_x_ - params_.min_vel_x_) * (std::sqrt(dist_goal_sq) / dist_thres);
    vy = 0.0;
    vt = params_.min_vel_th_ + (params_.max_vel_th_ - params_.min_vel_th_) * std::fabs(dth) / M_PI;
    if (dth < 0.0) vt *= -1;
    Trajectory t;
    if (scoreTrajectory(rx, ry, rt, rvx, rvy, rvt, vx, vy, vt, params_.max_trans_acc_, 0.0,
                        params_.max_rot_acc_, wpx, wpy, agents, t) != -1) {
      last_branch_ = kApproach;
      last_approach_traj_ = t;
      setCmd(cmd_vel, vx, vy, vt);
      return true;
    }
  }

  // ---- the (v,w) grid: ref :338-417 on the device --------------------------
  uploadWorld(agents);
  const sfw_robot_state rs{rx, ry, rt, rvx, rvy, rvt};
  const sfw_goal_args ga{params_.max_trans_acc_, 0.0, params_.max_rot_acc_, wpx, wpy};
  last_costs_.assign(linvels_.size() * angvels_.size(), SFW_COST_INVALID);
  const int32_t nv = static_cast<int32_t>(linvels_.size()), nw = static_cast<int32_t>(angvels_.size());
  const int rc = multi_ ? sfw_multi_score_grid(multi_, &rs, linvels_.data(), nv, angvels_.data(), nw, &ga,
                                               last_costs_.data(), &last_best_)
                        : sfw_score_grid(handle_, &rs, linvels_.data(), nv, angvels_.data(), nw, &ga, last_costs_.data(),
                                         &last_best_);
  if (rc != SFW_OK) raise(multi_ ? "sfw_multi_score_grid" : "sfw_score_grid", rc);
  grid_staged_ = true;
  if (last_best_.index >= 0) {  // ref :426-455
    last_branch_ = kGrid;
    setCmd(cmd_vel, last_best_.vx, 0.0, last_best_.vtheta);
    return true;
  }
  last_branch_ = kGridFailed;   // ref :456-468: stop the robot
  setCmd(cmd_vel, 0.0, 0.0, 0.0);
  return false;
}

bool SFWPlanner::getTrajectoryPoints(int64_t index, Trajectory &out) {
  if (!grid_staged_) return false;
  int S = static_cast<int>(params_.sim_time_ / params_.sim_granularity_ + 0.5);
  if (S == 0) S = 1;
  std::vector<double> pts(static_cast<size_t>(3) * S);
  int32_t n = 0;
  if ((multi_ ? sfw_multi_grid_points(multi_, index, pts.data(), S, &n) : sfw_grid_points(handle_, index, pts.data(), S, &n)) !=
      SFW_OK)
    return false;
  out.resetPoints();
  const int64_t nw = static_cast<int64_t>(angvels_.size());
  out.xv_ = linvels_[static_cast<size_t>(index / nw)];
  out.yv_ = 0.0;
  out.thetav_ = angvels_[static_cast<size_t>(index % nw)];
  out.cost_ = last_costs_.empty() ? -1.0 : last_costs_[static_cast<size_t>(index)];
  for (int i = 0; i < n; ++i) out.addPoint(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  return true;
}

bool SFWPlanner::getTrajectories(std::vector<Trajectory> &out) {
  if (!grid_staged_) return false;
  int S = static_cast<int>(params_.sim_time_ / params_.sim_granularity_ + 0.5);
  if (S == 0) S = 1;
  const int64_t nw = static_cast<int64_t>(angvels_.size());
  const int64_t T = static_cast<int64_t>(linvels_.size()) * nw;
  std::vector<double> pts(static_cast<size_t>(3) * S * T);
  std::vector<int32_t> n(static_cast<size_t>(T));
  if (multi_) {  // every rank dumps its own block of rows
    const int32_t R = sfw_multi_ranks(multi_);
    for (int32_t r = 0; r < R; ++r) {
      int32_t first = 0, rows = 0;
      if (sfw_multi_rank_rows(multi_, r, &first, &rows) != SFW_OK) return false;
      const int64_t lo = static_cast<int64_t>(first) * nw, cnt = static_cast<int64_t>(rows) * nw;
      if (cnt > 0 && sfw_grid_points_batch(sfw_multi_rank_handle(multi_, r), 0, cnt,
                                           pts.data() + static_cast<size_t>(lo) * 3 * S, n.data() + lo) != SFW_OK)
        return false;
    }
  } else if (sfw_grid_points_batch(handle_, 0, T, pts.data(), n.data()) != SFW_OK) {
    return false;
  }
  out.assign(static_cast<size_t>(T), Trajectory());
  for (int64_t i = 0; i < T; ++i) {
    Trajectory &t = out[static_cast<size_t>(i)];
    t.xv_ = linvels_[static_cast<size_t>(i / nw)];
    t.thetav_ = angvels_[static_cast<size_t>(i % nw)];
    t.cost_ = last_costs_[static_cast<size_t>(i)];
    const double *p = pts.data() + static_cast<size_t>(i) * 3 * S;
    for (int k = 0; k < n[static_cast<size_t>(i)]; ++k) t.addPoint(p[3 * k], p[3 * k + 1], p[3 * k + 2]);
  }
  return true;
}

bool SFWPlanner::getMarkers(std::vector<MarkerData> &out) {
  const size_t T = linvels_.size() * angvels_.size();
  if (T == 0) return false;  // empty sample sets: no markers to touch
  auto fill = [](MarkerData &m, const Trajectory &t, double z) {
    for (unsigned k = 0; k < t.getPointsSize(); ++k) {
      double x, y, th;
      t.getPoint(k, x, y, th);
      m.points.push_back(Point{x, y, z});
    }
  };
  if (last_branch_ == kApproach) {  // ref :309-325
    out.assign(T, MarkerData());
    for (size_t i = 0; i < T; ++i) out[i].id = static_cast<int>(i);
    fill(out[0], last_approach_traj_, 0.0);
    out[0].r = 0.0f; out[0].g = 1.0f; out[0].b = 0.0f; out[0].a = 1.0f;
    return true;
  }
  if (last_branch_ != kGrid && last_branch_ != kGridFailed) return false;
  std::vector<Trajectory> ts;
  if (!getTrajectories(ts)) return false;
  out.assign(T, MarkerData());
  for (size_t i = 0; i < T; ++i) {
    MarkerData &m = out[i];
    m.id = static_cast<int>(i);
    if (ts[i].xv_ == 0.0 && ts[i].thetav_ == 0.0) continue;  // ref :349-352: cleared, never scored
    const bool best = last_branch_ == kGrid && static_cast<int64_t>(i) == last_best_.index;
    fill(m, ts[i], best ? 0.1 : 0.0);                         // ref :366-374, :435-437
    if (best) { m.r = 0.0f; m.g = 1.0f; m.b = 0.0f; m.a = 1.0f; }          // ref :438-441
    else if (ts[i].cost_ < 0.0) { m.r = 1.0f; m.g = 0.0f; m.b = 0.0f; m.a = 0.6f; }  // ref :376-380
    else { m.r = 0.0f; m.g = 0.0f; m.b = 1.0f; m.a = 0.6f; }              // ref :381-385
  }
  return true;
}

bool SFWPlanner::updatePlan(const std::vector<PoseStamped> &new_plan) {  // ref :853-892
  goal_reached_ = false;
  global_plan_ = new_plan;
  if (global_plan_.empty()) {
    running_ = false;
    wp_index_ = -1;
    return true;
  }
  wp_index_ = 0;
  running_ = true;
  new_plan_ = true;
  const PoseStamped &goal = global_plan_.back();
  goal_x_ = goal.pose.position.x;
  goal_y_ = goal.pose.position.y;
  goal_t_ = getYaw(goal.pose.orientation);
  const PoseStamped &start = global_plan_.front();
  start_x_ = start.pose.position.x;
  start_y_ = start.pose.position.y;
  start_t_ = getYaw(start.pose.orientation);
  return true;
}

bool SFWPlanner::isGoalReached() {  // ref :894-900: one-shot flag
  if (goal_reached_) {
    goal_reached_ = false;
    return true;
  }
  return goal_reached_;
}

void SFWPlanner::resetGoal() { goal_reached_ = false; }

}  // namespace social_force_window_planner
