// sfw_planner_node.cpp — see sfw_planner_node.hpp.  Type-checked against tests/nav2_stubs/, never run here (needs ROS 2 Foxy + nav2).
// Control flow = reference src/sfw_planner_node.cpp:47-336; message <-> POD conversions only.
#include "sfw_planner_node.hpp"

#include <tf2/utils.h>

#include "nav2_core/exceptions.hpp"
#include "nav2_util/node_utils.hpp"
#include "pluginlib/class_list_macros.hpp"
#include "tf2_geometry_msgs/tf2_geometry_msgs.h"

namespace social_force_window_planner {

namespace {
template <class T>
T param(rclcpp_lifecycle::LifecycleNode *n, const std::string &key, const T &def) {
  nav2_util::declare_parameter_if_not_declared(n, key, rclcpp::ParameterValue(def));
  T v = def;
  n->get_parameter(key, v);
  return v;
}
PoseStamped toPod(const geometry_msgs::msg::PoseStamped &m) {
  PoseStamped p;
  p.frame_id = m.header.frame_id;
  p.stamp = rclcpp::Time(m.header.stamp).seconds();
  p.pose.position = Point{m.pose.position.x, m.pose.position.y, m.pose.position.z};
  p.pose.orientation = Quaternion{m.pose.orientation.x, m.pose.orientation.y, m.pose.orientation.z, m.pose.orientation.w};
  return p;
}
geometry_msgs::msg::PoseStamped toMsg(const PoseStamped &p, const std::string &frame, const rclcpp::Time &stamp) {
  geometry_msgs::msg::PoseStamped m;
  m.header.frame_id = frame;
  m.header.stamp = stamp;
  m.pose.position.x = p.pose.position.x;
  m.pose.position.y = p.pose.position.y;
  m.pose.orientation.x = p.pose.orientation.x;
  m.pose.orientation.y = p.pose.orientation.y;
  m.pose.orientation.z = p.pose.orientation.z;
  m.pose.orientation.w = p.pose.orientation.w;
  return m;
}
}  // namespace

// Parameter names exactly as reference sfw_planner.hpp:75-173.
ControllerParams SFWPlannerNode::readControllerParams() {
  auto *n = node_.get();
  const std::string p = name_ + ".";
  ControllerParams c;
  c.controller_frame_ = param<std::string>(n, p + "controller_frame", "odom");
  c.robot_base_frame_ = param<std::string>(n, p + "robot_base_frame", "base_link");
  c.max_vel_x_ = param(n, p + "max_trans_vel", 0.7);
  c.min_vel_x_ = param(n, p + "min_trans_vel", 0.1);
  c.max_vel_th_ = param(n, p + "max_rot_vel", 0.5);
  c.min_vel_th_ = param(n, p + "min_rot_vel", 0.1);
  c.max_trans_acc_ = param(n, p + "max_trans_acc", 1.0);
  c.max_rot_acc_ = param(n, p + "max_rot_acc", 1.0);
  c.min_in_place_vel_th_ = param(n, p + "min_in_place_rot_vel", 0.3);
  c.yaw_goal_tolerance_ = param(n, p + "yaw_goal_tolerance", 0.05);
  c.xy_goal_tolerance_ = param(n, p + "xy_goal_tolerance", 0.10);
  c.wp_tolerance_ = param(n, p + "wp_tolerance", 0.5);
  c.sim_time_ = param(n, p + "sim_time", 1.0);
  c.sim_granularity_ = param(n, p + "sim_granularity", 0.025);
  c.robot_radius_ = static_cast<float>(param(n, p + "robot_radius", 0.35));
  c.is_circular_ = param(n, p + "is_circular", true);
  c.social_weight_ = param(n, p + "social_weight", 1.2);
  c.costmap_weight_ = param(n, p + "costmap_weight", 2.0);
  c.angle_weight_ = param(n, p + "angle_weight", 0.7);
  c.distance_weight_ = param(n, p + "distance_weight", 1.0);
  c.vel_weight_ = param(n, p + "velocity_weight", 1.0);
  // not in the reference: arithmetic mode of the device kernels (include/sfw_hip.h) — "f64" (default), "f64_strict" (the pair
  // term's polynomials one degree longer), "f32" (forces in float)
  const std::string precision = param<std::string>(n, p + "device_precision", "f64");
  c.precision_ = precision == "f32" ? SFW_PRECISION_F32 : precision == "f64_strict" ? SFW_PRECISION_F64_STRICT : SFW_PRECISION_F64;
  if (precision != "f64" && precision != "f32" && precision != "f64_strict")
    RCLCPP_WARN(node_->get_logger(), "%sdevice_precision: unknown value '%s' (f64, f64_strict, f32): using f64", p.c_str(), precision.c_str());
  // Declared by the reference (sfw_planner.hpp:122-125, :136-138, :149-156) and read by nothing in it — scoreTrajectory takes
  // the step count from sim_granularity alone (:517 is a comment), the people's radius from the sensor interface's
  // person_radius, the force factors from lightsfm's own defaults — so they are declared here too, for a yaml written for the
  // reference (config/local_planner.yaml:20-26) to load without "undeclared parameter" complaints, and likewise ignored.
  (void)param(n, p + "angular_sim_granularity", 0.025);
  (void)param(n, p + "people_radius", 0.35);
  (void)param(n, p + "sfm_goal_weight", 2.0);
  (void)param(n, p + "sfm_obstacle_weight", 20.0);
  (void)param(n, p + "sfm_people_weight", 12.0);
  return c;
}
// Parameter names exactly as reference sensor_interface.hpp:78-124.
InterfaceParams SFWPlannerNode::readInterfaceParams() {
  auto *n = node_.get();
  const std::string p = name_ + ".", q = name_ + ".sensor_interface.";
  InterfaceParams i;
  i.max_robot_vel_x_ = static_cast<float>(param(n, p + "max_trans_vel", 0.7));
  i.robot_radius_ = static_cast<float>(param(n, p + "robot_radius", 0.35));
  i.person_radius_ = static_cast<float>(param(n, p + "person_radius", 0.35));
  i.robot_frame_ = param<std::string>(n, p + "robot_base_frame", "base_link");
  i.controller_frame_ = param<std::string>(n, p + "controller_frame", "odom");
  i.max_obstacle_dist_ = static_cast<float>(param(n, q + "max_obstacle_dist", 3.0));
  i.naive_goal_time_ = static_cast<float>(param(n, q + "naive_goal_time", 2.0));
  i.people_velocity_ = static_cast<float>(param(n, q + "people_velocity", 1.0));
  return i;
}

CostmapView SFWPlannerNode::costmapView() const {
  nav2_costmap_2d::Costmap2D *c = costmap_ros_->getCostmap();
  return CostmapView{c->getCharMap(), c->getSizeInCellsX(), c->getSizeInCellsY(), c->getOriginX(), c->getOriginY(),
                     c->getResolution()};
}

void SFWPlannerNode::configure(const rclcpp_lifecycle::LifecycleNode::SharedPtr &parent, std::string name,
                               const std::shared_ptr<tf2_ros::Buffer> &tf,
                               const std::shared_ptr<nav2_costmap_2d::Costmap2DROS> &costmap_ros) {
  node_ = parent;
  name_ = name;
  tf_ = tf;
  costmap_ros_ = costmap_ros;
  logger_ = parent->get_logger();
  const InterfaceParams ip = readInterfaceParams();
  // tf2 lookup reduced to the planar transform the sensor interface needs
  sensor_iface_ = std::make_shared<SFMSensorInterface>(ip, [this](const std::string &from, const std::string &to,
                                                                    Transform2D &out) {
    try {
      const auto t = tf_->lookupTransform(to, from, tf2::TimePointZero);
      out.tx = t.transform.translation.x;
      out.ty = t.transform.translation.y;
      out.yaw = tf2::getYaw(t.transform.rotation);
      return true;
    } catch (tf2::TransformException &) {
      return false;
    }
  });
  const std::string q = name_ + ".sensor_interface.";
  laser_sub_ = parent->create_subscription<sensor_msgs::msg::LaserScan>(
      param<std::string>(parent.get(), q + "laser_topic", "scan"), rclcpp::SensorDataQoS(),
      [this](const sensor_msgs::msg::LaserScan::SharedPtr m) {
        LaserScan s;
        s.frame_id = m->header.frame_id;
        s.angle_min = m->angle_min;
        s.angle_increment = m->angle_increment;
        s.ranges = m->ranges;
        sensor_iface_->laserCb(s);
      });
  people_sub_ = parent->create_subscription<people_msgs::msg::People>(
      param<std::string>(parent.get(), q + "people_topic", "people"), rclcpp::SensorDataQoS(),
      [this](const people_msgs::msg::People::SharedPtr m) {
        People pp;
        pp.frame_id = m->header.frame_id;
        for (const auto &pm : m->people) {
          Person p;
          p.position = Point{pm.position.x, pm.position.y, pm.position.z};
          p.velocity = Vector3{pm.velocity.x, pm.velocity.y, pm.velocity.z};
          p.tags = pm.tags;
          pp.people.push_back(p);
        }
        sensor_iface_->peopleCb(pp);
      });
  odom_sub_ = parent->create_subscription<nav_msgs::msg::Odometry>(
      param<std::string>(parent.get(), q + "odom_topic", "odom"), rclcpp::SensorDataQoS(),
      [this](const nav_msgs::msg::Odometry::SharedPtr m) {
        Odometry o;
        o.frame_id = m->header.frame_id;
        o.pose.position = Point{m->pose.pose.position.x, m->pose.pose.position.y, 0.0};
        o.pose.orientation = Quaternion{m->pose.pose.orientation.x, m->pose.pose.orientation.y,
                                        m->pose.pose.orientation.z, m->pose.pose.orientation.w};
        o.twist.linear = Vector3{m->twist.twist.linear.x, m->twist.twist.linear.y, 0.0};
        o.twist.angular.z = m->twist.twist.angular.z;
        sensor_iface_->odomCb(o);
      });
  global_path_pub_ = parent->create_publisher<nav_msgs::msg::Path>("robot_global_plan", 1);
  local_path_pub_ = parent->create_publisher<nav_msgs::msg::Path>("robot_local_plan", 1);
  traj_pub_ = parent->create_publisher<visualization_msgs::msg::MarkerArray>("robot_local_trajectories", 1);
  std::vector<Point> fp;
  for (const auto &q2 : costmap_ros_->getRobotFootprint()) fp.push_back(Point{q2.x, q2.y, 0.0});
  sfw_planner_ = std::make_shared<SFWPlanner>(readControllerParams(), sensor_iface_, costmapView(), fp, /*device*/ 0);
}

void SFWPlannerNode::cleanup() {
  global_path_pub_.reset();
  local_path_pub_.reset();
  traj_pub_.reset();
}
void SFWPlannerNode::activate() {
  global_path_pub_->on_activate();
  local_path_pub_->on_activate();
  traj_pub_->on_activate();
}
void SFWPlannerNode::deactivate() {
  global_path_pub_->on_deactivate();
  local_path_pub_->on_deactivate();
  traj_pub_->on_deactivate();
  sensor_iface_->stop();
}
void SFWPlannerNode::setPlan(const nav_msgs::msg::Path &path) {  // ref :114-117
  sensor_iface_->start();
  global_plan_ = path;
}

bool SFWPlannerNode::transformPose(const std::string &frame, const geometry_msgs::msg::PoseStamped &in,
                                   geometry_msgs::msg::PoseStamped &out) const {  // ref :187-204
  if (in.header.frame_id == frame) {
    out = in;
    return true;
  }
  try {
    tf_->transform(in, out, frame, tf2::durationFromSec(0.2));
    out.header.frame_id = frame;
    return true;
  } catch (tf2::TransformException &ex) {
    RCLCPP_ERROR(logger_, "Exception in transformPose: %s", ex.what());
  }
  return false;
}

geometry_msgs::msg::TwistStamped SFWPlannerNode::computeVelocityCommands(const geometry_msgs::msg::PoseStamped &pose,
                                                                         const geometry_msgs::msg::Twist &speed) {
  geometry_msgs::msg::TwistStamped vel;
  sensor_iface_->start();  // ref :226
  const std::string gframe = costmap_ros_->getGlobalFrameID();
  geometry_msgs::msg::PoseStamped robot_pose;
  if (!transformPose(gframe, pose, robot_pose))  // ref :235-238
    throw nav2_core::PlannerException("Unable to transform robot pose into costmap's frame");
  geometry_msgs::msg::PoseStamped robot_in_plan;
  if (global_plan_.poses.empty()) throw nav2_core::PlannerException("Received plan with zero length");
  if (!transformPose(global_plan_.header.frame_id, pose, robot_in_plan))  // ref :125-129
    throw nav2_core::PlannerException("Unable to transform robot pose into global plan's frame");
  std::vector<PoseStamped> plan;
  for (const auto &p : global_plan_.poses) plan.push_back(toPod(p));
  std::vector<PoseStamped> local;
  try {
    nav2_costmap_2d::Costmap2D *c = costmap_ros_->getCostmap();
    local = transformGlobalPlan(plan, toPod(robot_in_plan), c->getSizeInCellsX(), c->getSizeInCellsY(),
                                c->getResolution(), [&](const PoseStamped &in, PoseStamped &out) {
                                  geometry_msgs::msg::PoseStamped m =
                                      toMsg(in, global_plan_.header.frame_id, robot_in_plan.header.stamp), t;
                                  if (!transformPose(gframe, m, t)) return false;
                                  out = toPod(t);
                                  return true;
                                });
  } catch (const PlannerException &e) {
    throw nav2_core::PlannerException(e.what());
  }
  global_plan_.poses.erase(global_plan_.poses.begin(),
                           global_plan_.poses.begin() + (global_plan_.poses.size() - plan.size()));  // ref :176
  nav_msgs::msg::Path transformed;
  transformed.header.frame_id = gframe;
  transformed.header.stamp = robot_in_plan.header.stamp;
  for (const auto &p : local) transformed.poses.push_back(toMsg(p, gframe, robot_in_plan.header.stamp));
  global_path_pub_->publish(transformed);

  sfw_planner_->setParams(readControllerParams());  // the reference re-reads its parameters every cycle (:125)
  sfw_planner_->setCostmap(costmapView());
  sfw_planner_->updatePlan(local);                   // ref :277
  Twist cmd;
  Twist in_speed;
  in_speed.linear = Vector3{speed.linear.x, speed.linear.y, 0.0};
  in_speed.angular.z = speed.angular.z;
  // The reference fills and publishes its MarkerArray every cycle (:283, :301-309).  Here the points cost a device-to-host
  // copy (and, on grids too large for the capture, a second rollout): only while somebody listens to the topic.
  const bool want_markers = traj_pub_->get_subscription_count() > 0;
  sfw_planner_->setMarkerCapture(want_markers);
  const bool ok = sfw_planner_->findBestAction(toPod(robot_pose), in_speed, cmd);  // ref :281
  // the trajectory markers go out on both exits (red rejected, blue valid, green selected)
  std::vector<SFWPlanner::MarkerData> md;
  if (want_markers && sfw_planner_->getMarkers(md)) {
    visualization_msgs::msg::MarkerArray markers;
    const auto stamp = node_->get_clock()->now();
    for (const SFWPlanner::MarkerData &d : md) {
      visualization_msgs::msg::Marker m;               // ref src/sfw_planner.cpp:92-106
      m.header.frame_id = sfw_planner_->params().controller_frame_;
      m.header.stamp = stamp;
      m.ns = "trajectories";
      m.id = d.id;
      m.type = 4;                                      // LINE_STRIP
      m.action = 0;
      m.lifetime = rclcpp::Duration::from_seconds(0.3);
      m.scale.x = 0.01;
      m.color.r = d.r; m.color.g = d.g; m.color.b = d.b; m.color.a = d.a;
      m.pose.orientation.w = 1.0;
      for (const Point &q : d.points) {
        geometry_msgs::msg::Point gp;
        gp.x = q.x; gp.y = q.y; gp.z = q.z;
        m.points.push_back(gp);
      }
      markers.markers.push_back(m);
    }
    traj_pub_->publish(markers);
  }
  if (!ok) return vel;                               // ref :295-304: zero TwistStamped
  vel.header.stamp = node_->get_clock()->now();
  vel.header.frame_id = gframe;
  vel.twist.linear.x = cmd.linear.x;
  vel.twist.linear.y = cmd.linear.y;
  vel.twist.angular.z = cmd.angular.z;
  return vel;
}

}  // namespace social_force_window_planner

PLUGINLIB_EXPORT_CLASS(social_force_window_planner::SFWPlannerNode, nav2_core::Controller)
