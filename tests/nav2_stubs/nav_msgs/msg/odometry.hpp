#pragma once
#include <memory>
#include "geometry_msgs/msg/twist_stamped.hpp"
namespace nav_msgs { namespace msg {
struct PoseWithCovariance { geometry_msgs::msg::Pose pose; };
struct TwistWithCovariance { geometry_msgs::msg::Twist twist; };
struct Odometry { using SharedPtr = std::shared_ptr<Odometry>; std_msgs::msg::Header header; std::string child_frame_id;
                  PoseWithCovariance pose; TwistWithCovariance twist; };
} }
