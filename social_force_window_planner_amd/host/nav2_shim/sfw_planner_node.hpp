// sfw_planner_node.hpp — nav2 plugin shell over the MI355X scorer.
//
// NOT BUILT OR RUN IN THIS REPOSITORY'S IMAGE (no ROS 2 / nav2 there; see
// CMakeLists.txt: the target is only created when nav2_core is found).  It is
// type-checked on every test run against declarations of the names it uses
// (tests/nav2_stubs/, tests/test_nav2_shim_syntax.py: g++ -fsyntax-only -Werror).  It
// mirrors the reference class social_force_window_planner::SFWPlannerNode
// (reference include/social_force_window_planner/sfw_planner_node.hpp:53-175,
// src/sfw_planner_node.cpp:47-336): same class name, same base, same Foxy-era
// nav2_core::Controller signatures, same ROS parameter names, so an existing
// nav2 yaml (reference config/local_planner.yaml) and sfw_plugin.xml load
// unchanged.  Everything it calls is the ROS-free, tested code of ../.
#ifndef SFW_NAV2_SHIM_PLANNER_NODE_HPP_
#define SFW_NAV2_SHIM_PLANNER_NODE_HPP_

#include <memory>
#include <string>
#include <vector>

#include "geometry_msgs/msg/pose_stamped.hpp"
#include "geometry_msgs/msg/twist_stamped.hpp"
#include "nav2_core/controller.hpp"
#include "nav2_costmap_2d/costmap_2d_ros.hpp"
#include "nav_msgs/msg/odometry.hpp"
#include "nav_msgs/msg/path.hpp"
#include "people_msgs/msg/people.hpp"
#include "rclcpp/rclcpp.hpp"
#include "rclcpp_lifecycle/lifecycle_node.hpp"
#include "sensor_msgs/msg/laser_scan.hpp"
#include "tf2_ros/buffer.h"
#include "visualization_msgs/msg/marker_array.hpp"

#include "../plan_utils.hpp"
#include "../sensor_interface.hpp"
#include "../sfw_planner.hpp"

namespace social_force_window_planner {

class SFWPlannerNode : public nav2_core::Controller {
 public:
  SFWPlannerNode() = default;
  ~SFWPlannerNode() override = default;

  void configure(const rclcpp_lifecycle::LifecycleNode::SharedPtr &parent, std::string name,
                 const std::shared_ptr<tf2_ros::Buffer> &tf,
                 const std::shared_ptr<nav2_costmap_2d::Costmap2DROS> &costmap_ros) override;
  void cleanup() override;
  void activate() override;
  void deactivate() override;
  geometry_msgs::msg::TwistStamped computeVelocityCommands(const geometry_msgs::msg::PoseStamped &pose,
                                                           const geometry_msgs::msg::Twist &velocity) override;
  void setPlan(const nav_msgs::msg::Path &path) override;

 protected:
  bool transformPose(const std::string &frame, const geometry_msgs::msg::PoseStamped &in,
                     geometry_msgs::msg::PoseStamped &out) const;
  ControllerParams readControllerParams();
  InterfaceParams readInterfaceParams();
  CostmapView costmapView() const;

  rclcpp_lifecycle::LifecycleNode::SharedPtr node_;
  std::string name_;
  std::shared_ptr<tf2_ros::Buffer> tf_;
  std::shared_ptr<nav2_costmap_2d::Costmap2DROS> costmap_ros_;
  std::shared_ptr<SFWPlanner> sfw_planner_;
  std::shared_ptr<SFMSensorInterface> sensor_iface_;
  nav_msgs::msg::Path global_plan_;
  rclcpp::Logger logger_{rclcpp::get_logger("SFWPlanner")};
  rclcpp::Subscription<sensor_msgs::msg::LaserScan>::SharedPtr laser_sub_;
  rclcpp::Subscription<people_msgs::msg::People>::SharedPtr people_sub_;
  rclcpp::Subscription<nav_msgs::msg::Odometry>::SharedPtr odom_sub_;
  std::shared_ptr<rclcpp_lifecycle::LifecyclePublisher<nav_msgs::msg::Path>> global_path_pub_, local_path_pub_;
  std::shared_ptr<rclcpp_lifecycle::LifecyclePublisher<visualization_msgs::msg::MarkerArray>> traj_pub_;
};

}  // namespace social_force_window_planner
#endif
