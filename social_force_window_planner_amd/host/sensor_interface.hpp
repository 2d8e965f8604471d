// sensor_interface.hpp — ROS-free mirror of the reference's SFMSensorInterface
// (reference include/social_force_window_planner/sensor_interface.hpp:157-281,
// src/sensor_interface.cpp).  It turns laser / people / odometry inputs into the
// agent set the planner consumes: same class name, same
// callback names, same parameter names; ROS messages and tf2 are replaced by
// the PODs and the 2-D transform callback below.
#ifndef SFW_HOST_SENSOR_INTERFACE_HPP_
#define SFW_HOST_SENSOR_INTERFACE_HPP_

#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "sfw_planner.hpp"

namespace social_force_window_planner {

// reference sensor_interface.hpp:60-155 (numeric members are float there too)
struct InterfaceParams {
  float max_robot_vel_x_ = 0.7f, robot_radius_ = 0.35f, person_radius_ = 0.35f;
  std::string robot_frame_ = "base_link", controller_frame_ = "odom";
  float max_obstacle_dist_ = 3.0f, naive_goal_time_ = 2.0f, people_velocity_ = 1.0f;
};

// sensor_msgs/LaserScan subset
struct LaserScan {
  std::string frame_id;
  float angle_min = 0.0f, angle_increment = 0.0f;
  std::vector<float> ranges;
};
// people_msgs/Person subset: position.z carries the yaw, velocity.z the yaw rate,
// tags[0] = id, tags[1] = group id (reference src/sensor_interface.cpp:448-449,457,487)
struct Person {
  Point position;
  Vector3 velocity;
  std::vector<std::string> tags;
};
struct People {
  std::string frame_id;
  std::vector<Person> people;
};
// nav_msgs/Odometry subset
struct Odometry {
  std::string frame_id;
  Pose pose;
  Twist twist;
};
// Rigid 2-D transform source -> controller frame (stands in for tf2_ros::Buffer).
struct Transform2D {
  double tx = 0.0, ty = 0.0, yaw = 0.0;
};
using TransformLookup = std::function<bool(const std::string &from, const std::string &to, Transform2D &out)>;

class SFMSensorInterface : public AgentSource {
 public:
  explicit SFMSensorInterface(const InterfaceParams &params, TransformLookup tf = TransformLookup());

  void laserCb(const LaserScan &laser);     // reference src/sensor_interface.cpp:103-294
  void peopleCb(const People &people);      // :418-528
  void odomCb(const Odometry &odom);        // :534-581
  AgentSet getAgents() override;            // :618-631
  void start() { running_ = true; }         // sensor_interface.hpp:212
  void stop() { running_ = false; }
  void setParams(const InterfaceParams &p) { iface_params_ = p; }
  const std::vector<double> &obstacles() const { return obstacles_; }

 private:
  bool lookup(const std::string &from, Transform2D &t) const;
  InterfaceParams iface_params_;
  TransformLookup tf_;
  std::vector<sfw_agent> agents_;    // 0: robot, 1..: others
  std::vector<double> obstacles_;    // x,y pairs in the controller frame
  std::vector<double> agent_obstacles_;  // what the agents carry (assigned only in peopleCb, :513-524)
  People people_;
  std::mutex agents_mutex_, obs_mutex_, people_mutex_, odom_mutex_;
  bool running_ = false;
  bool laser_received_ = false;
  // left uninitialised by the reference constructor (sensor_interface.hpp:266);
  // false here, i.e. callbacks are ignored until the first odometry arrives
  bool odom_received_ = false;
};

}  // namespace social_force_window_planner
#endif
