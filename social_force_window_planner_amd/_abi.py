"""ctypes mirror of include/sfw_hip.h (POD structs + status codes).

Shared by the product binding (planner.py) and by the oracle binding under
oracle/ — the structs are the ABI, not an implementation.
"""
import ctypes as C

SFW_OK = 0
SFW_ERR_INVALID_ARG = -1
SFW_ERR_NO_DEVICE = -2
SFW_ERR_HIP = -3
SFW_ERR_STATE = -4
SFW_ERR_UNSUPPORTED = -5

SFW_COST_INVALID = -1.0
SFW_COST_SKIPPED = -2.0

SFW_PRECISION_F64 = 0
SFW_PRECISION_F32 = 1
SFW_PRECISION_F64_STRICT = 2

SFW_K2_AUTO, SFW_K2_REGISTER, SFW_K2_FLAT = -1, 0, 1
SFW_ORG_NONE, SFW_ORG_REGISTER_1, SFW_ORG_REGISTER_2, SFW_ORG_FLAT = 0, 1, 2, 3


class SfwParams(C.Structure):
    _fields_ = [
        ("max_vel_x", C.c_double),
        ("sim_time", C.c_double),
        ("sim_granularity", C.c_double),
        ("robot_radius", C.c_float),
        ("reserved0", C.c_float),
        ("social_weight", C.c_double),
        ("costmap_weight", C.c_double),
        ("angle_weight", C.c_double),
        ("distance_weight", C.c_double),
        ("vel_weight", C.c_double),
        ("robot_goal_radius", C.c_double),
        ("sfm_force_factor_desired", C.c_double),
        ("sfm_force_factor_obstacle", C.c_double),
        ("sfm_force_sigma_obstacle", C.c_double),
        ("sfm_force_factor_social", C.c_double),
        ("sfm_lambda", C.c_double),
        ("sfm_gamma", C.c_double),
        ("sfm_n", C.c_double),
        ("sfm_n_prime", C.c_double),
        ("sfm_relaxation_time", C.c_double),
        ("sfm_force_factor_group_gaze", C.c_double),
        ("sfm_force_factor_group_coherence", C.c_double),
        ("sfm_force_factor_group_repulsion", C.c_double),
        ("precision", C.c_int32),
        ("reserved1", C.c_int32),
    ]


def default_params(**overrides):
    """ControllerParams defaults (reference sfw_planner.hpp:56-66) plus
    lightsfm's sfm::Parameters defaults."""
    p = SfwParams()
    p.max_vel_x = 0.7
    p.sim_time = 1.0
    p.sim_granularity = 0.025
    p.robot_radius = 0.35
    p.social_weight = 1.2
    p.costmap_weight = 2.0
    p.angle_weight = 0.7
    p.distance_weight = 1.0
    p.vel_weight = 1.0
    p.robot_goal_radius = 0.20
    p.sfm_force_factor_desired = 2.0
    p.sfm_force_factor_obstacle = 10.0
    p.sfm_force_sigma_obstacle = 0.2
    p.sfm_force_factor_social = 2.1
    p.sfm_lambda = 2.0
    p.sfm_gamma = 0.35
    p.sfm_n = 2.0
    p.sfm_n_prime = 3.0
    p.sfm_relaxation_time = 0.5
    p.sfm_force_factor_group_gaze = 3.0
    p.sfm_force_factor_group_coherence = 2.0
    p.sfm_force_factor_group_repulsion = 1.0
    p.precision = SFW_PRECISION_F64
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"sfw_params has no field {k!r}")
        setattr(p, k, v)
    return p


class SfwAgent(C.Structure):
    _fields_ = [
        ("x", C.c_double),
        ("y", C.c_double),
        ("vx", C.c_double),
        ("vy", C.c_double),
        ("goal_x", C.c_double),
        ("goal_y", C.c_double),
        ("goal_radius", C.c_double),
        ("desired_velocity", C.c_double),
        ("radius", C.c_double),
        ("has_goal", C.c_int32),
        ("id", C.c_int32),
        ("group_id", C.c_int32),
        ("reserved", C.c_int32),
    ]


class SfwRobotState(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("x", "y", "theta", "vx", "vy", "vtheta")]


class SfwGoalArgs(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("acc_x", "acc_y", "acc_theta", "wpx", "wpy")]


class SfwBest(C.Structure):
    _fields_ = [
        ("index", C.c_int64),
        ("cost", C.c_double),
        ("vx", C.c_double),
        ("vy", C.c_double),
        ("vtheta", C.c_double),
        ("n_valid", C.c_int64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SfwBestKey(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("cost", "neg_linvel", "abs_angvel", "neg_index")]

    def as_tuple(self):
        return (self.cost, self.neg_linvel, self.abs_angvel, self.neg_index)


class SfwPlanInfo(C.Structure):
    _fields_ = [("split_step", C.c_int32), ("levels", C.c_int32), ("chunks", C.c_int32), ("organisation", C.c_int32),
                ("classes", C.c_int64), ("class_steps", C.c_int64), ("samples", C.c_int64), ("flat_samples", C.c_int64),
                ("one_launch", C.c_int64), ("rest_noise_unreproduced", C.c_int64)]

    def as_dict(self):
        return {"split_step": self.split_step, "levels": self.levels, "chunks": self.chunks, "classes": self.classes,
                "class_steps": self.class_steps, "samples": self.samples, "organisation": self.organisation,
                "flat_samples": self.flat_samples, "one_launch": self.one_launch,
                "rest_noise_unreproduced": self.rest_noise_unreproduced}


# Every symbol include/sfw_hip.h declares (checked by tests/test_abi.py).
EXPORTED_SYMBOLS = (
    "sfw_params_default",
    "sfw_abi_version",
    "sfw_create",
    "sfw_destroy",
    "sfw_set_params",
    "sfw_last_error",
    "sfw_set_costmap",
    "sfw_set_footprint",
    "sfw_set_agents",
    "sfw_score_grid",
    "sfw_score_one",
    "sfw_grid_stage",
    "sfw_grid_launch",
    "sfw_grid_sync",
    "sfw_grid_fetch",
    "sfw_grid_costs_view",
    "sfw_grid_plan_info",
    "sfw_plan_shared_prefix",
    "sfw_plan_row_blocks",
    "sfw_plan_axis_classes",
    "sfw_set_k2_form",
    "sfw_set_timing",
    "sfw_last_launch_ms",
    "sfw_last_clock_ghz",
    "sfw_grid_points",
    "sfw_grid_points_batch",
    "sfw_set_points_capture",
    "sfw_stream",
    "sfw_multi_create",
    "sfw_multi_destroy",
    "sfw_multi_last_error",
    "sfw_multi_describe",
    "sfw_multi_ranks",
    "sfw_multi_rank_handle",
    "sfw_multi_set_params",
    "sfw_multi_set_costmap",
    "sfw_multi_set_footprint",
    "sfw_multi_set_agents",
    "sfw_multi_score_grid",
    "sfw_multi_rank_rows",
    "sfw_multi_last_us",
    "sfw_multi_grid_points",
)
SFW_MULTI_RCCL, SFW_MULTI_HOST_REDUCE = 0, 1


class CtrlParams(C.Structure):
    """Numeric ControllerParams (reference sfw_planner.hpp:186-226), the layout of
    sfwh_params in host/sfw_host_capi.cpp."""

    _fields_ = [(n, C.c_double) for n in (
        "max_vel_x", "min_vel_x", "max_vel_th", "min_vel_th", "max_trans_acc", "max_rot_acc",
        "min_in_place_vel_th", "yaw_goal_tolerance", "xy_goal_tolerance", "wp_tolerance", "sim_time",
        "sim_granularity", "robot_radius", "social_weight", "costmap_weight", "angle_weight",
        "distance_weight", "vel_weight")] + [("is_circular", C.c_int32), ("precision", C.c_int32)]


def default_ctrl_params(**overrides):
    """ControllerParams() defaults, reference sfw_planner.hpp:56-66."""
    c = CtrlParams()
    c.max_vel_x, c.min_vel_x = 0.7, 0.1
    c.max_vel_th, c.min_vel_th = 0.5, 0.1
    c.max_trans_acc, c.max_rot_acc = 1.0, 1.0
    c.min_in_place_vel_th = 0.3
    c.yaw_goal_tolerance, c.xy_goal_tolerance, c.wp_tolerance = 0.05, 0.1, 0.5
    c.sim_time, c.sim_granularity = 1.0, 0.025
    c.robot_radius = float(C.c_float(0.35).value)
    c.social_weight, c.costmap_weight, c.angle_weight, c.distance_weight, c.vel_weight = 1.2, 2.0, 0.7, 1.0, 1.0
    c.is_circular, c.precision = 1, SFW_PRECISION_F64
    for k, v in overrides.items():
        if not hasattr(c, k):
            raise AttributeError(f"ControllerParams has no field {k!r}")
        setattr(c, k, v)
    return c


# SFWPlanner::Branch in host/sfw_planner.hpp (which exit of findBestAction ran)
BRANCH_NOT_RUNNING, BRANCH_GOAL_REACHED, BRANCH_ROTATE_IN_PLACE, BRANCH_ROTATE_BLOCKED = 0, 1, 2, 3
BRANCH_APPROACH, BRANCH_GRID, BRANCH_GRID_FAILED = 4, 5, 6
