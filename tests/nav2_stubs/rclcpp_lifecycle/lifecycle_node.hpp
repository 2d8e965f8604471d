#pragma once
#include <functional>
#include "rclcpp/rclcpp.hpp"
namespace rclcpp_lifecycle {
template <class M> class LifecyclePublisher {
 public:
  void publish(const M &) {}
  size_t get_subscription_count() const { return 0; }  // rclcpp::PublisherBase
  void on_activate() {}
  void on_deactivate() {}
};
class LifecycleNode {
 public:
  using SharedPtr = std::shared_ptr<LifecycleNode>;
  rclcpp::Logger get_logger() const { return rclcpp::Logger(); }
  rclcpp::Clock::SharedPtr get_clock() { return std::make_shared<rclcpp::Clock>(); }
  template <class T> bool get_parameter(const std::string &, T &) const { return false; }
  bool has_parameter(const std::string &) const { return false; }
  void declare_parameter(const std::string &, const rclcpp::ParameterValue &) {}
  template <class M, class CB>
  typename rclcpp::Subscription<M>::SharedPtr create_subscription(const std::string &, const rclcpp::QoS &, CB &&cb) {
    std::function<void(const typename M::SharedPtr)> f = cb;  // the callback must accept the message's SharedPtr
    (void)f;
    return nullptr;
  }
  template <class M> std::shared_ptr<LifecyclePublisher<M>> create_publisher(const std::string &, int) { return nullptr; }
};
}  // namespace rclcpp_lifecycle
