#include "plan_utils.hpp"

#include <algorithm>
#include <cmath>

namespace social_force_window_planner {

namespace {
inline double euclidean(const PoseStamped &a, const PoseStamped &b) {  // nav2_util::geometry_utils::euclidean_distance
  return std::hypot(a.pose.position.x - b.pose.position.x, a.pose.position.y - b.pose.position.y);
}
}  // namespace

std::vector<PoseStamped> transformGlobalPlan(std::vector<PoseStamped> &global_plan, const PoseStamped &robot_pose,
                                             unsigned size_x_cells, unsigned size_y_cells, double resolution,
                                             const PoseTransform &to_costmap) {
  if (global_plan.empty()) throw PlannerException("Received plan with zero length");  // ref :121-123
  const double max_costmap_dim = std::max(size_x_cells, size_y_cells);                 // ref :131-135
  const double max_transform_dist = max_costmap_dim * resolution / 2.0;
  // first pose of minimal distance (the reference's min_by keeps the first minimum, :28-44)
  auto begin = global_plan.begin();
  double lowest = euclidean(robot_pose, *begin);
  for (auto it = global_plan.begin() + 1; it != global_plan.end(); ++it) {
    const double d = euclidean(robot_pose, *it);
    if (d < lowest) { lowest = d; begin = it; }
  }
  auto end = std::find_if(begin, global_plan.end(), [&](const PoseStamped &p) {      // ref :143-148
    return euclidean(robot_pose, p) > max_transform_dist;
  });
  std::vector<PoseStamped> out;
  for (auto it = begin; it != end; ++it) {                                             // ref :150-164
    PoseStamped stamped = *it, transformed;
    stamped.stamp = robot_pose.stamp;
    if (!to_costmap || !to_costmap(stamped, transformed)) transformed = PoseStamped();  // the reference ignores the failure
    out.push_back(transformed);
  }
  global_plan.erase(global_plan.begin(), begin);                                       // ref :176
  if (out.empty()) throw PlannerException("Resulting plan has 0 poses in it.");        // ref :179-181
  return out;
}

}  // namespace social_force_window_planner
