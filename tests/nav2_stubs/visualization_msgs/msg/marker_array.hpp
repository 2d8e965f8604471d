#pragma once
#include <string>
#include <vector>
#include "builtin_interfaces/msg/time.hpp"
#include "geometry_msgs/msg/pose_stamped.hpp"
namespace builtin_interfaces { namespace msg { struct Duration { int32_t sec = 0; uint32_t nanosec = 0; }; } }
namespace std_msgs { namespace msg { struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; } }
namespace visualization_msgs { namespace msg {
struct Marker {
  std_msgs::msg::Header header;
  std::string ns;
  int32_t id = 0, type = 0, action = 0;
  geometry_msgs::msg::Pose pose;
  geometry_msgs::msg::Vector3 scale;
  std_msgs::msg::ColorRGBA color;
  builtin_interfaces::msg::Duration lifetime;
  std::vector<geometry_msgs::msg::Point> points;
};
struct MarkerArray { std::vector<Marker> markers; };
} }
