"""C++ host mirror of the reference's SFWPlanner (ROS-free) over the C ABI,
plus a ctypes binding of its flat test shim (sfw_host_capi.cpp)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .._abi import CtrlParams, default_ctrl_params

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsfw_host.so")
_lib = None


def build(force=False):
    r = subprocess.run(["make", "-C", _HERE, "all"] + (["-B"] if force else []), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libsfw_host.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.sfwh_params_default.argtypes = [C.POINTER(CtrlParams)]
        L.sfwh_params_default.restype = None
        L.sfwh_create.argtypes = [C.POINTER(CtrlParams), vp, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
                                  C.c_double, vp, C.c_int32, C.c_int32]
        L.sfwh_create.restype = vp
        L.sfwh_destroy.argtypes = [vp]
        L.sfwh_destroy.restype = None
        L.sfwh_last_error.argtypes = [vp]
        L.sfwh_last_error.restype = C.c_char_p
        L.sfwh_set_params.argtypes = [vp, C.POINTER(CtrlParams)]
        L.sfwh_set_costmap.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double]
        L.sfwh_set_agents.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
        L.sfwh_set_sample_sets.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
        L.sfwh_set_devices.argtypes = [vp, vp, C.c_int32, C.c_int32]
        L.sfwh_ranks.argtypes = [vp]
        L.sfwh_update_plan.argtypes = [vp, vp, C.c_int32]
        L.sfwh_find_best_action.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.sfwh_is_goal_reached.argtypes = [vp]
        L.sfwh_wp_index.argtypes = [vp]
        L.sfwh_running.argtypes = [vp]
        L.sfwh_last_costs.argtypes = [vp, vp, C.c_int64]
        L.sfwh_last_costs.restype = C.c_int64
        L.sfwh_trajectory_points.argtypes = [vp, C.c_int64, vp, C.c_int32]
        L.sfwh_all_trajectories.argtypes = [vp, vp, C.c_int32, vp]
        L.sfwh_all_trajectories.restype = C.c_int64
        L.sfwh_markers.argtypes = [vp, vp, vp, vp, C.c_int64]
        L.sfwh_markers.restype = C.c_int64
        L.sfwh_set_marker_capture.argtypes = [vp, C.c_int32]
        L.sfwh_set_marker_capture.restype = None
        L.sfwh_get_yaw.argtypes = [C.c_double] * 4
        L.sfwh_get_yaw.restype = C.c_double
        _lib = L
    return _lib


class HostPlanner:
    """Drives the C++ SFWPlanner (host/sfw_planner.hpp) the way nav2's
    controller_server drives the reference: updatePlan, then findBestAction per
    control cycle."""

    def __init__(self, ctrl: CtrlParams | None, scene, device=0):
        self.ctrl = ctrl if ctrl is not None else default_ctrl_params()
        cells = np.ascontiguousarray(scene.cells, dtype=np.uint8)
        fp = np.ascontiguousarray(scene.footprint, dtype=np.float64).reshape(-1, 2)
        sy, sx = cells.shape
        self._h = lib().sfwh_create(C.byref(self.ctrl), cells.ctypes.data, sx, sy, scene.origin_x,
                                    scene.origin_y, scene.resolution, fp.ctypes.data if len(fp) else None,
                                    len(fp), device)
        self.set_agents(scene.agents, scene.obstacles)

    def close(self):
        if getattr(self, "_h", None):
            lib().sfwh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what}: {(lib().sfwh_last_error(self._h) or b'').decode()}")
        return rc

    def set_params(self, ctrl):
        self.ctrl = ctrl
        self._check(lib().sfwh_set_params(self._h, C.byref(ctrl)), "set_params")

    def set_agents(self, agents, obstacles=None):
        obs = np.ascontiguousarray(obstacles if obstacles is not None else np.zeros((0, 2)), dtype=np.float64)
        obs = obs.reshape(-1, 2)
        n = len(agents)
        lib().sfwh_set_agents(self._h, C.addressof(agents) if n else None, n,
                              obs.ctypes.data if len(obs) else None, len(obs))

    def set_costmap(self, cells, ox, oy, res):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        sy, sx = cells.shape
        lib().sfwh_set_costmap(self._h, cells.ctypes.data, sx, sy, ox, oy, res)

    def markers(self, n_samples):
        """SFWPlanner::getMarkers: (rgba[T,4], point counts[T], z of the first point[T]) or None."""
        rgba = np.zeros((n_samples, 4), dtype=np.float32)
        counts = np.zeros(n_samples, dtype=np.int32)
        z0 = np.zeros(n_samples, dtype=np.float64)
        n = lib().sfwh_markers(self._h, rgba.ctypes.data, counts.ctypes.data, z0.ctypes.data, n_samples)
        if n > n_samples:
            raise ValueError(f"markers({n_samples}): the planner holds {n} samples")
        return None if n < 0 else (rgba, counts, z0)

    def set_marker_capture(self, on=True):
        """SFWPlanner::setMarkerCapture: the scoring launch also leaves the Trajectory points (one copy per marker dump)."""
        lib().sfwh_set_marker_capture(self._h, 1 if on else 0)

    def set_devices(self, devices, host_reduce=False):
        """SFWPlanner::setDevices: grid rows over several devices from this process (before the first scoring call)."""
        d = (C.c_int * len(devices))(*devices)
        self._check(lib().sfwh_set_devices(self._h, d, len(devices), 1 if host_reduce else 0), "setDevices")

    def ranks(self):
        return lib().sfwh_ranks(self._h)

    def set_sample_sets(self, lin, ang):
        lin = np.ascontiguousarray(lin, dtype=np.float64)
        ang = np.ascontiguousarray(ang, dtype=np.float64)
        lib().sfwh_set_sample_sets(self._h, lin.ctypes.data, len(lin), ang.ctypes.data, len(ang))

    def update_plan(self, xyyaw):
        p = np.ascontiguousarray(xyyaw, dtype=np.float64).reshape(-1, 3)
        self._check(lib().sfwh_update_plan(self._h, p.ctypes.data if len(p) else None, len(p)), "updatePlan")

    def find_best_action(self, pose, vel):
        pose = np.ascontiguousarray(pose, dtype=np.float64)
        vel = np.ascontiguousarray(vel, dtype=np.float64)
        cmd = np.zeros(3, dtype=np.float64)
        found, branch = C.c_int32(), C.c_int32()
        self._check(lib().sfwh_find_best_action(self._h, pose.ctypes.data, vel.ctypes.data, cmd.ctypes.data,
                                                C.byref(found), C.byref(branch)), "findBestAction")
        return bool(found.value), cmd, branch.value

    def is_goal_reached(self):
        return bool(lib().sfwh_is_goal_reached(self._h))

    @property
    def wp_index(self):
        return lib().sfwh_wp_index(self._h)

    @property
    def running(self):
        return bool(lib().sfwh_running(self._h))

    def last_costs(self):
        n = lib().sfwh_last_costs(self._h, None, 0)
        out = np.zeros(n, dtype=np.float64)
        lib().sfwh_last_costs(self._h, out.ctypes.data, n)
        return out

    def all_trajectories(self, n_samples, cap):
        pts = np.zeros((n_samples, cap, 3), dtype=np.float64)
        counts = np.zeros(n_samples, dtype=np.int32)
        n = lib().sfwh_all_trajectories(self._h, pts.ctypes.data, cap, counts.ctypes.data)
        if n < 0:
            raise RuntimeError("getTrajectories failed")
        return pts[:n], counts[:n]

    def trajectory_points(self, index, cap=4096):
        pts = np.zeros((cap, 3), dtype=np.float64)
        n = self._check(lib().sfwh_trajectory_points(self._h, index, pts.ctypes.data, cap), "getTrajectoryPoints")
        return pts[:n].copy()
