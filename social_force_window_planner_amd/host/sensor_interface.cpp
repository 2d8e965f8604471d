// sensor_interface.cpp — see sensor_interface.hpp.  Reference:
// src/sensor_interface.cpp (laserCb :103-294, peopleCb :418-528, odomCb :534-581,
// getAgents :618-631).
#include "sensor_interface.hpp"

#include <cmath>

namespace social_force_window_planner {

SFMSensorInterface::SFMSensorInterface(const InterfaceParams &params, TransformLookup tf)
    : iface_params_(params), tf_(std::move(tf)) {
  // ref :31-37: one agent, the robot
  sfw_agent robot{};
  robot.desired_velocity = iface_params_.max_robot_vel_x_;
  robot.radius = iface_params_.robot_radius_;
  robot.has_goal = 0;
  robot.group_id = -1;
  // The reference never sets the robot's id while computeSocialWork's single-agent force
  // skips a person whose id equals the robot's (ref :692-699): a value no tracker tag can take keeps
  // every person's Wp term (a person tagged "0" would lose it with id 0).
  robot.id = SFW_ROBOT_ID_NONE;
  agents_.assign(1, robot);
}

bool SFMSensorInterface::lookup(const std::string &from, Transform2D &t) const {
  if (from == iface_params_.controller_frame_) {
    t = Transform2D{};
    return true;
  }
  return tf_ ? tf_(from, iface_params_.controller_frame_, t) : false;
}

void SFMSensorInterface::laserCb(const LaserScan &laser) {
  if (!running_ || !odom_received_) return;  // ref :106-107
  laser_received_ = true;
  std::vector<double> points;  // x,y pairs
  float angle = laser.angle_min;
  for (size_t i = 0; i < laser.ranges.size(); ++i) {  // ref :120-131, float arithmetic
    const float r = laser.ranges[i];
    if (!std::isnan(r) && std::isfinite(r) && r < iface_params_.max_obstacle_dist_) {
      points.push_back(r * std::cos(angle));
      points.push_back(r * std::sin(angle));
    }
    angle += laser.angle_increment;
  }
  if (points.empty()) {  // ref :133-139
    std::lock_guard<std::mutex> l(obs_mutex_);
    obstacles_.clear();
    return;
  }
  if (laser.frame_id != iface_params_.controller_frame_) {  // ref :143-169
    Transform2D t;
    if (lookup(laser.frame_id, t)) {
      const double c = std::cos(t.yaw), s = std::sin(t.yaw);
      for (size_t i = 0; i + 1 < points.size(); i += 2) {
        const double x = points[i], y = points[i + 1];
        points[i] = t.tx + c * x - s * y;
        points[i + 1] = t.ty + s * x + c * y;
      }
    }  // on failure the reference keeps the untransformed point (catch ... continue)
  }
  People people;
  {
    std::lock_guard<std::mutex> l(people_mutex_);
    people = people_;
  }
  std::vector<Point> people_points;  // ref :181-208
  if (!people.people.empty()) {
    Transform2D t;
    const bool same = people.frame_id == iface_params_.controller_frame_;
    if (!same && !lookup(people.frame_id, t)) return;  // ref :195-202
    const double c = std::cos(t.yaw), s = std::sin(t.yaw);
    for (const Person &p : people.people) {
      Point q = p.position;
      if (!same) {
        q.x = t.tx + c * p.position.x - s * p.position.y;
        q.y = t.ty + s * p.position.x + c * p.position.y;
      }
      people_points.push_back(q);
    }
  }
  if (!people_points.empty()) {  // ref :211-229: drop points on a person (float hypot, <=)
    std::vector<double> kept;
    for (size_t i = 0; i + 1 < points.size(); i += 2) {
      bool remove = false;
      for (const Point &person : people_points) {
        const float dx = static_cast<float>(points[i] - person.x);
        const float dy = static_cast<float>(points[i + 1] - person.y);
        if (std::hypot(dx, dy) <= iface_params_.person_radius_) {
          remove = true;
          break;
        }
      }
      if (!remove) {
        kept.push_back(points[i]);
        kept.push_back(points[i + 1]);
      }
    }
    points.swap(kept);
  }
  std::lock_guard<std::mutex> l(obs_mutex_);  // ref :288-290
  obstacles_ = points;
}

void SFMSensorInterface::peopleCb(const People &people) {
  if (!running_ || !odom_received_) return;  // ref :421-422
  {
    std::lock_guard<std::mutex> l(people_mutex_);
    people_ = people;
  }
  Transform2D t;
  const bool same = people.frame_id == iface_params_.controller_frame_;
  if (!same && !lookup(people.frame_id, t)) return;  // ref :464-468
  const double c = std::cos(t.yaw), s = std::sin(t.yaw);
  std::vector<sfw_agent> agents;
  for (const Person &p : people.people) {
    sfw_agent ag{};
    ag.id = p.tags.size() > 0 ? std::stoi(p.tags[0]) : 0;        // ref :448
    ag.group_id = p.tags.size() > 1 ? std::stoi(p.tags[1]) : -1;  // ref :449
    double px = p.position.x, py = p.position.y, yaw = p.position.z;  // ref :454-458
    double vx = p.velocity.x, vy = p.velocity.y;                      // ref :472-475
    if (!same) {
      px = t.tx + c * p.position.x - s * p.position.y;
      py = t.ty + s * p.position.x + c * p.position.y;
      yaw += t.yaw;
      vx = c * p.velocity.x - s * p.velocity.y;  // transformVector: rotation only (:640-669)
      vy = s * p.velocity.x + c * p.velocity.y;
    }
    ag.x = px;
    ag.y = py;
    ag.vx = vx;
    ag.vy = vy;
    // linearVelocity / yaw (:481-487) are not consumed by the scoring path
    ag.radius = iface_params_.person_radius_;
    // naive goal = pos + naive_goal_time * vel (:494-503)
    ag.goal_x = px + iface_params_.naive_goal_time_ * vx;
    ag.goal_y = py + iface_params_.naive_goal_time_ * vy;
    ag.goal_radius = iface_params_.person_radius_;
    ag.has_goal = 1;
    ag.desired_velocity = iface_params_.people_velocity_;
    (void)yaw;
    agents.push_back(ag);
  }
  std::vector<double> obs;  // ref :513-520
  {
    std::lock_guard<std::mutex> l(obs_mutex_);
    obs = obstacles_;
  }
  std::lock_guard<std::mutex> l(agents_mutex_);  // ref :522-527
  agents_.resize(people.people.size() + 1);
  agent_obstacles_ = obs;
  for (size_t i = 1; i < agents_.size(); ++i) agents_[i] = agents[i - 1];
}

void SFMSensorInterface::odomCb(const Odometry &odom) {
  if (!running_) return;  // ref :536-537
  odom_received_ = true;
  std::lock_guard<std::mutex> l(agents_mutex_);
  sfw_agent &robot = agents_[0];  // ref :552-580
  robot.x = odom.pose.position.x;
  robot.y = odom.pose.position.y;
  // the twist of an odometry message is in the robot-local frame and is stored as is (:565-575)
  robot.vx = odom.twist.linear.x;
  robot.vy = odom.twist.linear.y;
}

AgentSet SFMSensorInterface::getAgents() {  // ref :618-631
  std::lock_guard<std::mutex> l(agents_mutex_);
  AgentSet out;
  out.agents = agents_;
  out.obstacles_xy = agent_obstacles_;
  return out;
}

}  // namespace social_force_window_planner
