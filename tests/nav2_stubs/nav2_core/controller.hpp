#pragma once
// nav2_core::Controller as of ROS 2 Foxy: the pure virtuals the reference plugin overrides
// (reference include/social_force_window_planner/sfw_planner_node.hpp:73-116).
#include <memory>
#include <string>
#include "geometry_msgs/msg/twist_stamped.hpp"
#include "nav2_costmap_2d/costmap_2d_ros.hpp"
#include "nav_msgs/msg/path.hpp"
#include "rclcpp_lifecycle/lifecycle_node.hpp"
#include "tf2_ros/buffer.h"
namespace nav2_core {
class Controller {
 public:
  using Ptr = std::shared_ptr<Controller>;
  virtual ~Controller() {}
  virtual void configure(const rclcpp_lifecycle::LifecycleNode::SharedPtr &, std::string name,
                         const std::shared_ptr<tf2_ros::Buffer> &, const std::shared_ptr<nav2_costmap_2d::Costmap2DROS> &) = 0;
  virtual void cleanup() = 0;
  virtual void activate() = 0;
  virtual void deactivate() = 0;
  virtual void setPlan(const nav_msgs::msg::Path &path) = 0;
  virtual geometry_msgs::msg::TwistStamped computeVelocityCommands(const geometry_msgs::msg::PoseStamped &pose,
                                                                   const geometry_msgs::msg::Twist &velocity) = 0;
};
}  // namespace nav2_core
