#pragma once
#include <memory>
#include <vector>
#include "std_msgs/msg/header.hpp"
namespace sensor_msgs { namespace msg {
struct LaserScan { using SharedPtr = std::shared_ptr<LaserScan>; std_msgs::msg::Header header;
                   float angle_min = 0, angle_max = 0, angle_increment = 0, range_min = 0, range_max = 0;
                   std::vector<float> ranges; };
} }
