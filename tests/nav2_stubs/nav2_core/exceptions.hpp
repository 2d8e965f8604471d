#pragma once
#include <stdexcept>
#include <string>
namespace nav2_core {
class PlannerException : public std::runtime_error {
 public:
  explicit PlannerException(const std::string &description) : std::runtime_error(description) {}
};
}  // namespace nav2_core
