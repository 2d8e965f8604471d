#pragma once
#include <string>
#include "rclcpp/rclcpp.hpp"
namespace nav2_util {
template <class NodeT> void declare_parameter_if_not_declared(NodeT node, const std::string &name, const rclcpp::ParameterValue &v) {
  if (!node->has_parameter(name)) node->declare_parameter(name, v);
}
}  // namespace nav2_util
