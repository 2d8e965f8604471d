// sfw_planner.hpp — ROS-free C++ host mirror of the reference planner core
// (reference include/social_force_window_planner/sfw_planner.hpp:234-476,
// src/sfw_planner.cpp).  Same class name, same public methods, same argument
// meaning and error behaviour; the (v,w) grid loop and the two single-sample
// scoreTrajectory call sites go through the C ABI (include/sfw_hip.h) to the
// MI355X kernels instead of the reference's serial CPU loop.
//
// What is different on purpose (DESIGN.md "boundary"):
//   * ROS message types are replaced by the small PODs below (same field names
//     as geometry_msgs), so this file compiles without ROS 2 / nav2.
//   * `const nav2_costmap_2d::Costmap2D&` becomes CostmapView (a borrowed
//     pointer + geometry); a snapshot is uploaded on every findBestAction.
//   * `std::shared_ptr<SFMSensorInterface>` becomes AgentSource (getAgents()).
//   * RViz markers become getTrajectoryPoints() (optional, off by default).
#ifndef SFW_HOST_PLANNER_HPP_
#define SFW_HOST_PLANNER_HPP_

#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sfw_hip.h"

namespace social_force_window_planner {

// ---- geometry_msgs stand-ins (field names as in ROS) ----------------------
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std::string frame_id; double stamp = 0; Pose pose; };
struct Twist { Vector3 linear, angular; };
double getYaw(const Quaternion &q);               // tf2::getYaw
Quaternion quaternionFromYaw(double yaw);

// ---- reference ControllerParams (sfw_planner.hpp:55-227), same names ------
struct ControllerParams {
  std::string controller_frame_ = "odom", robot_base_frame_ = "base_link";
  double max_vel_x_ = 0.7, min_vel_x_ = 0.1;
  double max_vel_th_ = 0.5, min_vel_th_ = 0.1;
  double max_trans_acc_ = 1.0, max_rot_acc_ = 1.0;
  double min_in_place_vel_th_ = 0.3;
  double yaw_goal_tolerance_ = 0.05, xy_goal_tolerance_ = 0.1, wp_tolerance_ = 0.5;
  double sim_time_ = 1.0, sim_granularity_ = 0.025, angular_sim_granularity_ = 0.025;
  float robot_radius_ = 0.35f, people_radius_ = 0.35f;
  bool is_circular_ = true;
  float sfm_goal_weight_ = 2.0f, sfm_obstacle_weight_ = 20.0f, sfm_people_weight_ = 12.0f;  // read, never applied
  double social_weight_ = 1.2, costmap_weight_ = 2.0, angle_weight_ = 0.7, distance_weight_ = 1.0,
         vel_weight_ = 1.0;
  // not in the reference: arithmetic mode of the device kernels (SFW_PRECISION_F64, _F64_STRICT — the same with longer
  // polynomials in the pair term —, or _F32: forces in float; include/sfw_hip.h)
  int precision_ = SFW_PRECISION_F64;
  sfw_params toAbi() const;
};

// Borrowed view of the live costmap (stands in for const Costmap2D&).
struct CostmapView {
  const uint8_t *cells = nullptr;  // row-major, y outer: cells[my*size_x+mx]
  uint32_t size_x = 0, size_y = 0;
  double origin_x = 0, origin_y = 0, resolution = 0.05;
};

// Stands in for SFMSensorInterface::getAgents() (sensor_interface.hpp:207-213).
struct AgentSet {
  std::vector<sfw_agent> agents;       // [0] = robot
  std::vector<double> obstacles_xy;    // shared laser points
};
class AgentSource {
 public:
  virtual ~AgentSource() = default;
  virtual AgentSet getAgents() = 0;
};

// Reference Trajectory container (trajectory.hpp:45-118) — same public API.
class Trajectory {
 public:
  Trajectory();
  Trajectory(double xv, double yv, double thetav, double time_delta, unsigned int num_pts);
  double xv_, yv_, thetav_;
  double cost_;
  double time_delta_ = 0.0;
  void getPoint(unsigned int index, double &x, double &y, double &th) const;
  void setPoint(unsigned int index, double x, double y, double th);
  void addPoint(double x, double y, double th);
  void getEndpoint(double &x, double &y, double &th) const;
  void resetPoints();
  unsigned int getPointsSize() const;

 private:
  struct P3 { double x, y, th; };
  std::vector<P3> pts_;
};

// The sample sets the reference builds in its constructor (src/sfw_planner.cpp:64-85).
void referenceSampleSets(double max_vel_x, double max_vel_th, std::vector<double> &linvels,
                         std::vector<double> &angvels);

class SFWPlanner {
 public:
  // device: HIP device ordinal.  The first call that needs scoring throws
  // std::runtime_error when the device library cannot be initialised (there is
  // no CPU scoring path).
  SFWPlanner(const ControllerParams &params, std::shared_ptr<AgentSource> sensor_iface,
             const CostmapView &costmap, std::vector<Point> footprint_spec, int device = 0);
  ~SFWPlanner();
  SFWPlanner(const SFWPlanner &) = delete;
  SFWPlanner &operator=(const SFWPlanner &) = delete;

  // reference sfw_planner.hpp:261-263.  false => no valid command (cmd_vel zeroed
  // or, in the rotate-in-place branch, the rejected rotation).
  bool findBestAction(const PoseStamped &global_pose, const Twist &global_vel, Twist &cmd_vel);
  bool updatePlan(const std::vector<PoseStamped> &new_plan);  // :271
  bool isGoalReached();                                       // :273
  void resetGoal();                                           // :274
  void setFootprint(std::vector<Point> footprint);            // :277
  std::vector<Point> getFootprint() const { return footprint_spec_; }

  // Parameters are re-read every cycle in the reference (src/sfw_planner.cpp:125).
  void setParams(const ControllerParams &p);
  const ControllerParams &params() const { return params_; }
  void setCostmap(const CostmapView &c) { costmap_ = c; }
  // Score the grid on several devices from this one process (the reference plugin is one process,
  // sfw_plugin.xml:1-9): rows of the grid split over the listed devices, one RCCL all-reduce(min) picks
  // the command (include/sfw_hip.h: sfw_multi_*).  Call before the first scoring call.  host_reduce: exchange
  // on the host instead of RCCL, which lets a device be listed more than once (tests on a one-GPU box).
  // The two single-sample branches (:204-206, :299-301) run on the first listed device.
  void setDevices(std::vector<int> devices, bool host_reduce = false);
  int ranks() const { return multi_ ? sfw_multi_ranks(multi_) : 1; }
  // Replace the sample sets (BASELINE.json grids); default = reference 5 x 9.
  void setSampleSets(std::vector<double> linvels, std::vector<double> angvels);
  const std::vector<double> &linvels() const { return linvels_; }
  const std::vector<double> &angvels() const { return angvels_; }

  // reference sfw_planner.hpp:309-314 (private there; public here for tests).
  double scoreTrajectory(double x, double y, double theta, double vx, double vy, double vtheta,
                         double vx_samp, double vy_samp, double vtheta_samp, double acc_x, double acc_y,
                         double acc_theta, double wpx, double wpy, const AgentSet &agents, Trajectory &traj);

  // Results of the last grid evaluation (replaces the MarkerArray).
  const std::vector<double> &lastCosts() const { return last_costs_; }
  const sfw_best &lastBest() const { return last_best_; }
  int lastBranch() const { return last_branch_; }  // see Branch
  enum Branch { kNotRunning = 0, kGoalReached, kRotateInPlace, kRotateBlocked, kApproach, kGrid, kGridFailed };
  // Trajectory points of sample `index` of the last grid (src/sfw_planner.cpp:366-374).
  bool getTrajectoryPoints(int64_t index, Trajectory &out);
  // All samples of the last grid in one device call = what the reference's MarkerArray holds
  // after the loop (:347-386): points of every scored sample, cost_ < 0 for rejected ones
  // (drawn red there), the winner is lastBest().index (drawn green).
  bool getTrajectories(std::vector<Trajectory> &out);
  // What the reference's MarkerArray holds after this cycle (getMarkers(), src/sfw_planner.cpp:86-113): one LINE_STRIP
  // marker per sample (ns "trajectories", id = iteration index, scale.x 0.01, lifetime 0.3 s, pose.orientation.w 1).
  // Grid branch (:347-386, :435-440): points of every scored sample at z = 0, red (1,0,0,0.6) when its cost is < 0, blue
  // (0,0,1,0.6) otherwise; the selected sample green (0,1,0,1) with its points raised to z = 0.1; the never-scored (0,0)
  // sample empty.  Approach branch (:309-325): marker 0 = the approach trajectory, green.  (The reference means to clear
  // the other markers there but iterates over copies, `for (auto m : markers_.markers)`, so they keep last cycle's points;
  // here they are cleared, as intended.)  false: the cycle took a branch that leaves the markers untouched.
  struct MarkerData {
    int id = 0;
    float r = 0, g = 0, b = 0, a = 1;
    std::vector<Point> points;
  };
  bool getMarkers(std::vector<MarkerData> &out);
  // Whether the cycle's scoring launch should also leave the Trajectory points (sfw_set_points_capture): on while
  // somebody listens to the markers — getMarkers is then one device-to-host copy per cycle —, off otherwise (the
  // reference fills its MarkerArray every cycle, :347-386; nothing needs the points when nobody subscribes).
  void setMarkerCapture(bool on);
  int wpIndex() const { return wp_index_; }
  bool running() const { return running_; }

 private:
  void ensureDevice();
  void uploadWorld(const AgentSet &agents);
  [[noreturn]] void raise(const char *what, int status) const;

  std::mutex configuration_mutex_;
  ControllerParams params_;
  std::shared_ptr<AgentSource> sensor_iface_;
  CostmapView costmap_;
  std::vector<Point> footprint_spec_;
  std::vector<PoseStamped> global_plan_;
  std::vector<double> linvels_, angvels_;
  sfw_handle handle_ = nullptr;      // single device; with multi_: rank 0's handle (owned by multi_)
  sfw_multi_handle multi_ = nullptr;
  std::vector<int> devices_;         // non-empty: multi-device mode
  bool host_reduce_ = false;
  int device_ = 0;
  std::vector<double> last_costs_;
  Trajectory last_approach_traj_;
  sfw_best last_best_{};
  int last_branch_ = kNotRunning;
  bool grid_staged_ = false;
  bool marker_capture_ = false;

  int wp_index_ = -1;
  bool running_ = false, new_plan_ = false, goal_reached_ = false;
  double start_x_ = 0, start_y_ = 0, start_t_ = 0, goal_x_ = 0, goal_y_ = 0, goal_t_ = 0;
};

}  // namespace social_force_window_planner
#endif
