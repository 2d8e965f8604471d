#pragma once
#include <chrono>
#include <stdexcept>
#include "geometry_msgs/msg/pose_stamped.hpp"
namespace tf2 {
using TimePoint = std::chrono::time_point<std::chrono::system_clock, std::chrono::nanoseconds>;
using Duration = std::chrono::nanoseconds;
const TimePoint TimePointZero = TimePoint(Duration::zero());
inline Duration durationFromSec(double) { return Duration::zero(); }
class TransformException : public std::runtime_error { public: explicit TransformException(const std::string &m) : std::runtime_error(m) {} };
inline double getYaw(const geometry_msgs::msg::Quaternion &) { return 0.0; }
}  // namespace tf2
