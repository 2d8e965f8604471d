#pragma once
#include "geometry_msgs/msg/pose_stamped.hpp"
namespace geometry_msgs { namespace msg {
struct Twist { Vector3 linear, angular; };
struct TwistStamped { std_msgs::msg::Header header; Twist twist; };
} }
