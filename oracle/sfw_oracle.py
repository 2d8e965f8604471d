"""ctypes binding of the CPU oracle (oracle/libsfw_oracle.so).

TEST INFRASTRUCTURE: import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Mirrors the product binding
(social_force_window_planner_amd/planner.py) method for method so parity
tests read symmetrically.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from social_force_window_planner_amd._abi import (
    SFW_OK,
    SfwAgent,
    SfwBest,
    SfwGoalArgs,
    SfwParams,
    SfwRobotState,
    default_params,
)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsfw_oracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libsfw_ref.so")
_lib = None
_ref = None


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    src = os.path.join(_HERE, "sfw_oracle.cpp")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale or (os.path.isdir("/root/reference") and not os.path.exists(_REF_PATH)):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.sfwo_create.argtypes = [C.POINTER(SfwParams), C.POINTER(C.c_void_p)]
        L.sfwo_destroy.argtypes = [C.c_void_p]
        L.sfwo_set_params.argtypes = [C.c_void_p, C.POINTER(SfwParams)]
        L.sfwo_set_costmap.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double,
                                       C.c_double, C.c_double]
        L.sfwo_set_footprint.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.sfwo_set_agents.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.sfwo_score_one.argtypes = [C.c_void_p, C.POINTER(SfwRobotState), C.c_double, C.c_double,
                                     C.c_double, C.POINTER(SfwGoalArgs), C.POINTER(C.c_double),
                                     C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.sfwo_score_grid.argtypes = [C.c_void_p, C.POINTER(SfwRobotState), C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_int32, C.POINTER(SfwGoalArgs), C.c_void_p,
                                      C.POINTER(SfwBest), C.c_int32]
        L.sfwo_select_best.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.POINTER(SfwBest)]
        L.sfwo_line_cells.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int32]
        L.sfwo_footprint_cost.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.sfwo_footprint_cost.restype = C.c_double
        L.sfwo_pair_force.argtypes = [C.POINTER(SfwParams), C.POINTER(SfwAgent), C.POINTER(SfwAgent),
                                      C.c_void_p]
        L.sfwo_normalize_angle.argtypes = [C.c_float] * 3
        L.sfwo_normalize_angle.restype = C.c_float
        L.sfwo_num_steps.argtypes = [C.POINTER(SfwParams)]
        L.sfwo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def ref_lib():
    """The real-reference shim (LineIterator + Trajectory), or None if it was
    never built (it is built in the authoring container, where /root/reference
    exists, and travels to the GPU box as a prebuilt .so)."""
    global _ref
    if _ref is None and os.path.exists(_REF_PATH):
        R = C.CDLL(_REF_PATH)
        R.ref_line_cells.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_int]
        R.ref_trajectory_roundtrip.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _ref = R
    return _ref


def _check(rc, what):
    if rc != SFW_OK:
        raise RuntimeError(f"oracle {what} failed: status {rc}")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleScorer:
    def __init__(self, params: SfwParams | None = None):
        self.params = params if params is not None else default_params()
        self._h = C.c_void_p()
        _check(lib().sfwo_create(C.byref(self.params), C.byref(self._h)), "create")

    def close(self):
        if self._h:
            lib().sfwo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        self.params = params
        _check(lib().sfwo_set_params(self._h, C.byref(params)), "set_params")

    def set_costmap(self, cells, origin_x, origin_y, resolution):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        sy, sx = cells.shape
        _check(lib().sfwo_set_costmap(self._h, cells.ctypes.data, sx, sy, origin_x, origin_y, resolution),
               "set_costmap")

    def set_footprint(self, xy):
        xy = _f64(xy).reshape(-1, 2)
        _check(lib().sfwo_set_footprint(self._h, xy.ctypes.data if len(xy) else None, len(xy)),
               "set_footprint")

    def set_agents(self, agents, obstacles=None):
        n = len(agents)
        obs = _f64(obstacles if obstacles is not None else np.zeros((0, 2))).reshape(-1, 2)
        rc = lib().sfwo_set_agents(self._h, C.addressof(agents) if n else None, n,
                                   obs.ctypes.data if len(obs) else None, len(obs))
        _check(rc, "set_agents")

    def load_scene(self, scene):
        self.set_costmap(scene.cells, scene.origin_x, scene.origin_y, scene.resolution)
        self.set_footprint(scene.footprint)
        self.set_agents(scene.agents, scene.obstacles)

    def score_grid(self, robot_state, linvels, angvels, goal_args, n_threads=1):
        lin, ang = _f64(linvels), _f64(angvels)
        rs, ga = SfwRobotState(*robot_state), SfwGoalArgs(*goal_args)
        costs = np.empty(len(lin) * len(ang), dtype=np.float64)
        best = SfwBest()
        _check(lib().sfwo_score_grid(self._h, C.byref(rs), lin.ctypes.data, len(lin), ang.ctypes.data,
                                     len(ang), C.byref(ga), costs.ctypes.data, C.byref(best), n_threads),
               "score_grid")
        return costs, best.as_dict()

    def score_one(self, robot_state, vx_samp, vy_samp, vth_samp, goal_args, points_cap=4096):
        rs, ga = SfwRobotState(*robot_state), SfwGoalArgs(*goal_args)
        cost = C.c_double()
        pts = np.zeros((points_cap, 3), dtype=np.float64)
        n = C.c_int32()
        _check(lib().sfwo_score_one(self._h, C.byref(rs), vx_samp, vy_samp, vth_samp, C.byref(ga),
                                    C.byref(cost), pts.ctypes.data, points_cap, C.byref(n)), "score_one")
        return cost.value, pts[: min(n.value, points_cap)].copy()

    def footprint_cost(self, x, y, theta):
        return lib().sfwo_footprint_cost(self._h, x, y, theta)


def select_best(linvels, angvels, costs):
    lin, ang, c = _f64(linvels), _f64(angvels), _f64(costs)
    best = SfwBest()
    _check(lib().sfwo_select_best(lin.ctypes.data, len(lin), ang.ctypes.data, len(ang), c.ctypes.data,
                                  C.byref(best)), "select_best")
    return best.as_dict()


def line_cells(x0, y0, x1, y1, use_ref=False):
    cap = max(abs(x1 - x0), abs(y1 - y0)) + 2
    out = np.zeros((cap, 2), dtype=np.int32)
    fn = ref_lib().ref_line_cells if use_ref else lib().sfwo_line_cells
    n = fn(x0, y0, x1, y1, out.ctypes.data, cap)
    return out[:n].copy()


def pair_force(params, me: SfwAgent, other: SfwAgent):
    out = np.zeros(2, dtype=np.float64)
    lib().sfwo_pair_force(C.byref(params), C.byref(me), C.byref(other), out.ctypes.data)
    return out


class OraclePlanner:
    """CPU restatement of findBestAction/updatePlan/isGoalReached around an
    OracleScorer (reference src/sfw_planner.cpp:117-468, :853-902).  Same
    surface as social_force_window_planner_amd.host.HostPlanner."""

    def __init__(self, ctrl, scene):
        from social_force_window_planner_amd._abi import CtrlParams, default_ctrl_params

        L = lib()
        L.sfwo_planner_create.argtypes = [C.c_void_p, C.POINTER(CtrlParams)]
        L.sfwo_planner_create.restype = C.c_void_p
        L.sfwo_planner_destroy.argtypes = [C.c_void_p]
        L.sfwo_planner_destroy.restype = None
        L.sfwo_planner_set_params.argtypes = [C.c_void_p, C.POINTER(CtrlParams)]
        L.sfwo_planner_set_params.restype = None
        L.sfwo_planner_set_sample_sets.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.sfwo_planner_set_sample_sets.restype = None
        L.sfwo_planner_update_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.sfwo_planner_update_plan.restype = None
        L.sfwo_planner_find_best_action.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.sfwo_planner_is_goal_reached.argtypes = [C.c_void_p]
        L.sfwo_planner_wp_index.argtypes = [C.c_void_p]
        L.sfwo_planner_running.argtypes = [C.c_void_p]
        L.sfwo_planner_last_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.sfwo_planner_last_costs.restype = C.c_int64
        self.ctrl = ctrl if ctrl is not None else default_ctrl_params()
        self.scorer = OracleScorer(default_params())
        self.scorer.load_scene(scene)
        self._p = C.c_void_p(L.sfwo_planner_create(self.scorer._h, C.byref(self.ctrl)))

    def close(self):
        if getattr(self, "_p", None):
            lib().sfwo_planner_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, ctrl):
        self.ctrl = ctrl
        lib().sfwo_planner_set_params(self._p, C.byref(ctrl))

    def set_agents(self, agents, obstacles=None):
        self.scorer.set_agents(agents, obstacles)

    def set_costmap(self, cells, ox, oy, res):
        self.scorer.set_costmap(cells, ox, oy, res)

    def set_sample_sets(self, lin, ang):
        lin, ang = _f64(lin), _f64(ang)
        lib().sfwo_planner_set_sample_sets(self._p, lin.ctypes.data, len(lin), ang.ctypes.data, len(ang))

    def update_plan(self, xyyaw):
        p = _f64(xyyaw).reshape(-1, 3)
        lib().sfwo_planner_update_plan(self._p, p.ctypes.data if len(p) else None, len(p))

    def find_best_action(self, pose, vel):
        pose, vel = _f64(pose), _f64(vel)
        cmd = np.zeros(3, dtype=np.float64)
        found, branch = C.c_int32(), C.c_int32()
        lib().sfwo_planner_find_best_action(self._p, pose.ctypes.data, vel.ctypes.data, cmd.ctypes.data,
                                            C.byref(found), C.byref(branch))
        return bool(found.value), cmd, branch.value

    def is_goal_reached(self):
        return bool(lib().sfwo_planner_is_goal_reached(self._p))

    @property
    def wp_index(self):
        return lib().sfwo_planner_wp_index(self._p)

    @property
    def running(self):
        return bool(lib().sfwo_planner_running(self._p))

    def last_costs(self):
        n = lib().sfwo_planner_last_costs(self._p, None, 0)
        out = np.zeros(n, dtype=np.float64)
        lib().sfwo_planner_last_costs(self._p, out.ctypes.data, n)
        return out
