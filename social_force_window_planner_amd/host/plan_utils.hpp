// plan_utils.hpp — ROS-free core of SFWPlannerNode::transformGlobalPlan
// (reference src/sfw_planner_node.cpp:119-185): prune the global plan to the part
// that starts at the pose closest to the robot and stays within half the
// costmap's larger dimension, transform it into the costmap frame, and drop the
// poses already behind the robot from the stored plan.
#ifndef SFW_HOST_PLAN_UTILS_HPP_
#define SFW_HOST_PLAN_UTILS_HPP_

#include <functional>
#include <stdexcept>
#include <vector>

#include "sfw_planner.hpp"

namespace social_force_window_planner {

// Stands in for nav2_core::PlannerException (same messages as the reference).
struct PlannerException : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// pose in the plan's frame -> pose in the costmap's global frame; false on failure
using PoseTransform = std::function<bool(const PoseStamped &in, PoseStamped &out)>;

// global_plan is modified in place (prefix erased, ref :176).  robot_pose_in_plan_frame
// = the robot pose already transformed into the plan's frame (ref :125-129).
std::vector<PoseStamped> transformGlobalPlan(std::vector<PoseStamped> &global_plan,
                                             const PoseStamped &robot_pose_in_plan_frame, unsigned size_x_cells,
                                             unsigned size_y_cells, double resolution, const PoseTransform &to_costmap);

}  // namespace social_force_window_planner
#endif
