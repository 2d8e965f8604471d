#pragma once
#include <memory>
#include <string>
#include <vector>
#include "builtin_interfaces/msg/time.hpp"
namespace rclcpp {
class Logger {};
inline Logger get_logger(const std::string &) { return Logger(); }
class Time {
 public:
  Time() = default;
  Time(const builtin_interfaces::msg::Time &) {}
  double seconds() const { return 0.0; }
  operator builtin_interfaces::msg::Time() const { return builtin_interfaces::msg::Time(); }
};
class Duration {
 public:
  static Duration from_seconds(double) { return Duration(); }
  template <class D> operator D() const { return D(); }  // -> builtin_interfaces::msg::Duration
};
class Clock { public: using SharedPtr = std::shared_ptr<Clock>; Time now() const { return Time(); } };
class ParameterValue {
 public:
  ParameterValue() = default;
  ParameterValue(bool) {}
  ParameterValue(int) {}
  ParameterValue(double) {}
  ParameterValue(const char *) {}
  ParameterValue(const std::string &) {}
};
class QoS { public: explicit QoS(int = 10) {} };
class SensorDataQoS : public QoS { public: SensorDataQoS() : QoS(5) {} };
template <class M> class Subscription { public: using SharedPtr = std::shared_ptr<Subscription<M>>; };
}  // namespace rclcpp
#define RCLCPP_ERROR(logger, ...) do { (void)(logger); } while (0)
#define RCLCPP_INFO(logger, ...) do { (void)(logger); } while (0)
#define RCLCPP_WARN(logger, ...) do { (void)(logger); } while (0)
