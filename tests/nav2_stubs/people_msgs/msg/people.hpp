#pragma once
#include <memory>
#include <string>
#include <vector>
#include "geometry_msgs/msg/pose_stamped.hpp"
namespace people_msgs { namespace msg {
struct Person { std::string name; geometry_msgs::msg::Point position, velocity; double reliability = 0;
                std::vector<std::string> tagnames, tags; };
struct People { using SharedPtr = std::shared_ptr<People>; std_msgs::msg::Header header; std::vector<Person> people; };
} }
