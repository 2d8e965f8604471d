"""ctypes binding of the product C-ABI library (libsfw_hip.so, include/sfw_hip.h).

This is plumbing for tests and bench.py; the C++ host mirror of the reference's
SFWPlanner lives in host/.  There is no CPU fallback here: if the HIP library
is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from ._abi import (
    EXPORTED_SYMBOLS,
    SFW_ERR_NO_DEVICE,
    SFW_OK,
    SfwAgent,
    SfwBest,
    SfwBestKey,
    SfwGoalArgs,
    SfwParams,
    SfwRobotState,
    SfwPlanInfo,
    default_params,
)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFW_HIP_LIB: load another build of the same library (A/B kernel tuning in one GPU session;
# boxes differ by ~10 % in sustained clocks, so variants are only comparable within one run)
LIB_PATH = os.environ.get("SFW_HIP_LIB") or os.path.join(_HERE, "libsfw_hip.so")
_lib = None


class SfwError(RuntimeError):
    def __init__(self, status, what, detail=""):
        super().__init__(f"{what} failed: status {status}" + (f" ({detail})" if detail else ""))
        self.status = status


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles on CPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in ("sfw_capi.hip", "sfw_kernels.hip", "sfw_kernels_strict.hip", "sfw_device.h", "sfw_math.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "sfw_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(LIB_PATH) < os.path.getmtime(s) for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-C", csrc, "all"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building libsfw_hip.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the scoring path)")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.sfw_params_default.argtypes = [C.POINTER(SfwParams)]
        L.sfw_params_default.restype = None
        L.sfw_abi_version.restype = C.c_int
        L.sfw_create.argtypes = [C.POINTER(SfwParams), C.c_int, C.POINTER(vp)]
        L.sfw_destroy.argtypes = [vp]
        L.sfw_set_params.argtypes = [vp, C.POINTER(SfwParams)]
        L.sfw_last_error.argtypes = [vp]
        L.sfw_last_error.restype = C.c_char_p
        L.sfw_set_costmap.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double]
        L.sfw_set_footprint.argtypes = [vp, vp, C.c_int32]
        L.sfw_set_agents.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
        L.sfw_score_grid.argtypes = [vp, C.POINTER(SfwRobotState), vp, C.c_int32, vp, C.c_int32,
                                     C.POINTER(SfwGoalArgs), vp, C.POINTER(SfwBest)]
        L.sfw_score_one.argtypes = [vp, C.POINTER(SfwRobotState), C.c_double, C.c_double, C.c_double,
                                    C.POINTER(SfwGoalArgs), C.POINTER(C.c_double), vp, C.c_int32,
                                    C.POINTER(C.c_int32)]
        L.sfw_grid_stage.argtypes = [vp, C.POINTER(SfwRobotState), vp, C.c_int32, vp, C.c_int32,
                                     C.POINTER(SfwGoalArgs), C.c_int64]
        L.sfw_grid_launch.argtypes = [vp]
        L.sfw_grid_sync.argtypes = [vp]
        L.sfw_grid_fetch.argtypes = [vp, vp, C.POINTER(SfwBest), C.POINTER(SfwBestKey)]
        L.sfw_grid_plan_info.argtypes = [vp, C.POINTER(SfwPlanInfo)]
        L.sfw_set_k2_form.argtypes = [vp, C.c_int32]
        L.sfw_set_timing.argtypes = [vp, C.c_int32]
        L.sfw_last_launch_ms.argtypes = [vp, C.c_int32, C.POINTER(C.c_float)]
        L.sfw_last_clock_ghz.argtypes = [vp, C.POINTER(C.c_double)]
        L.sfw_grid_points.argtypes = [vp, C.c_int64, vp, C.c_int32, C.POINTER(C.c_int32)]
        L.sfw_grid_points_batch.argtypes = [vp, C.c_int64, C.c_int64, vp, vp]
        L.sfw_set_points_capture.argtypes = [vp, C.c_int32]
        L.sfw_stream.argtypes = [vp]
        L.sfw_stream.restype = vp
        L.sfw_multi_create.argtypes = [C.POINTER(SfwParams), C.POINTER(C.c_int), C.c_int32, C.c_int32, C.POINTER(vp)]
        L.sfw_multi_destroy.argtypes = [vp]
        L.sfw_multi_last_error.argtypes = [vp]
        L.sfw_multi_last_error.restype = C.c_char_p
        L.sfw_multi_ranks.argtypes = [vp]
        L.sfw_multi_ranks.restype = C.c_int32
        L.sfw_multi_rank_handle.argtypes = [vp, C.c_int32]
        L.sfw_multi_rank_handle.restype = vp
        L.sfw_multi_set_params.argtypes = [vp, C.POINTER(SfwParams)]
        L.sfw_multi_set_costmap.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double]
        L.sfw_multi_set_footprint.argtypes = [vp, vp, C.c_int32]
        L.sfw_multi_set_agents.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
        L.sfw_multi_score_grid.argtypes = [vp, C.POINTER(SfwRobotState), vp, C.c_int32, vp, C.c_int32,
                                           C.POINTER(SfwGoalArgs), vp, C.POINTER(SfwBest)]
        L.sfw_multi_last_us.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
        L.sfw_multi_rank_rows.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.sfw_plan_row_blocks.argtypes = [vp, C.c_int32, vp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int32, C.c_int32, C.c_int32, vp]
        L.sfw_grid_costs_view.argtypes = [vp]
        L.sfw_grid_costs_view.restype = C.c_void_p
        L.sfw_plan_shared_prefix.argtypes = [vp, C.c_int32, vp, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                             C.c_double, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.POINTER(C.c_int32)]
        L.sfw_multi_grid_points.argtypes = [vp, C.c_int64, vp, C.c_int32, C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def exported_symbols():
    L = lib()
    return {name: hasattr(L, name) for name in EXPORTED_SYMBOLS}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class HipScorer:
    """The (v,w) grid scorer on one MI355X.  Method-for-method the same surface
    as the CPU checker's scorer class under oracle/ (which this package never imports)."""

    def __init__(self, params: SfwParams | None = None, device: int = 0):
        self.params = params if params is not None else default_params()
        self._h = C.c_void_p()
        rc = lib().sfw_create(C.byref(self.params), device, C.byref(self._h))
        if rc == SFW_ERR_NO_DEVICE:
            raise SfwError(rc, "sfw_create", "no HIP device visible; this library has no CPU fallback")
        if rc != SFW_OK:
            raise SfwError(rc, "sfw_create")
        self._grid = None

    def close(self):
        if getattr(self, "_h", None):
            lib().sfw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != SFW_OK:
            raise SfwError(rc, what, (lib().sfw_last_error(self._h) or b"").decode())

    # -- world state -------------------------------------------------------
    def set_params(self, params):
        self._check(lib().sfw_set_params(self._h, C.byref(params)), "sfw_set_params")
        self.params = params

    def set_costmap(self, cells, origin_x, origin_y, resolution):
        cells = np.ascontiguousarray(cells, dtype=np.uint8)
        sy, sx = cells.shape
        self._check(lib().sfw_set_costmap(self._h, cells.ctypes.data, sx, sy, origin_x, origin_y, resolution),
                    "sfw_set_costmap")

    def set_footprint(self, xy):
        xy = _f64(xy).reshape(-1, 2)
        self._check(lib().sfw_set_footprint(self._h, xy.ctypes.data if len(xy) else None, len(xy)),
                    "sfw_set_footprint")

    def set_agents(self, agents, obstacles=None):
        n = len(agents)
        obs = _f64(obstacles if obstacles is not None else np.zeros((0, 2))).reshape(-1, 2)
        self._check(lib().sfw_set_agents(self._h, C.addressof(agents) if n else None, n,
                                         obs.ctypes.data if len(obs) else None, len(obs)), "sfw_set_agents")

    def load_scene(self, scene):
        self.set_costmap(scene.cells, scene.origin_x, scene.origin_y, scene.resolution)
        self.set_footprint(scene.footprint)
        self.set_agents(scene.agents, scene.obstacles)

    # -- scoring -----------------------------------------------------------
    def score_grid(self, robot_state, linvels, angvels, goal_args):
        lin, ang = _f64(linvels), _f64(angvels)
        rs, ga = SfwRobotState(*robot_state), SfwGoalArgs(*goal_args)
        costs = np.empty(len(lin) * len(ang), dtype=np.float64)
        best = SfwBest()
        self._check(lib().sfw_score_grid(self._h, C.byref(rs), lin.ctypes.data, len(lin), ang.ctypes.data,
                                         len(ang), C.byref(ga), costs.ctypes.data, C.byref(best)),
                    "sfw_score_grid")
        self._grid = (len(lin), len(ang))
        return costs, best.as_dict()

    def score_one(self, robot_state, vx_samp, vy_samp, vth_samp, goal_args, points_cap=4096):
        rs, ga = SfwRobotState(*robot_state), SfwGoalArgs(*goal_args)
        cost = C.c_double()
        pts = np.zeros((points_cap, 3), dtype=np.float64)
        n = C.c_int32()
        self._check(lib().sfw_score_one(self._h, C.byref(rs), vx_samp, vy_samp, vth_samp, C.byref(ga),
                                        C.byref(cost), pts.ctypes.data, points_cap, C.byref(n)),
                    "sfw_score_one")
        self._grid = None
        return cost.value, pts[: min(n.value, points_cap)].copy()

    # -- device-resident pipeline -----------------------------------------
    def stage(self, robot_state, linvels, angvels, goal_args, index_base=0):
        lin, ang = _f64(linvels), _f64(angvels)
        rs, ga = SfwRobotState(*robot_state), SfwGoalArgs(*goal_args)
        self._check(lib().sfw_grid_stage(self._h, C.byref(rs), lin.ctypes.data, len(lin), ang.ctypes.data,
                                         len(ang), C.byref(ga), index_base), "sfw_grid_stage")
        self._grid = (len(lin), len(ang))

    def prepared(self, robot_state, linvels, angvels, goal_args, index_base=0, zero_copy=False):
        """The blocking call with its arguments marshalled ONCE (a C caller builds its structs once too): step() is
        sfw_grid_stage + sfw_grid_launch + sfw_grid_fetch into the same cost buffer every cycle, three foreign calls and
        nothing else — no numpy array, no ctypes struct is created per call."""
        return PreparedGrid(self, robot_state, linvels, angvels, goal_args, index_base, zero_copy)

    def launch(self):
        self._check(lib().sfw_grid_launch(self._h), "sfw_grid_launch")

    def sync(self):
        self._check(lib().sfw_grid_sync(self._h), "sfw_grid_sync")

    def fetch(self, want_costs=True, out=None):
        """out: a caller-owned float64 buffer of nv * nw doubles the cost vector is written to (a C caller passes the same
        array every cycle; a fresh np.empty of >= 128 KB is an mmap + page faults per call)."""
        nv, nw = self._grid
        if want_costs and out is not None:
            if out.dtype != np.float64 or out.size != nv * nw or not out.flags.c_contiguous:
                raise ValueError("fetch(out=...): need a contiguous float64 array of nv * nw elements")
            costs = out
        else:
            costs = np.empty(nv * nw, dtype=np.float64) if want_costs else None
        best, key = SfwBest(), SfwBestKey()
        self._check(lib().sfw_grid_fetch(self._h, costs.ctypes.data if want_costs else None, C.byref(best),
                                         C.byref(key)), "sfw_grid_fetch")
        return costs, best.as_dict(), key.as_tuple()

    def costs_view(self):
        """sfw_grid_costs_view as a read-only numpy array over the handle's pinned cost vector (valid until the next launch),
        or None when the last launch was not mirrored."""
        p = lib().sfw_grid_costs_view(self._h)
        if not p:
            return None
        nv, nw = self._grid
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(nv * nw,))
        a.flags.writeable = False
        return a

    def plan_info(self):
        """How the staged grid will be launched (shared-prefix split step, classes, chunks)."""
        info = SfwPlanInfo()
        self._check(lib().sfw_grid_plan_info(self._h, C.byref(info)), "sfw_grid_plan_info")
        return info.as_dict()

    def set_k2_form(self, form):
        """SFW_K2_AUTO / SFW_K2_REGISTER / SFW_K2_FLAT: the organisation of the social-force kernel's waves
        (bit-identical results; tests and tuning)."""
        self._check(lib().sfw_set_k2_form(self._h, form), "sfw_set_k2_form")

    def set_timing(self, enabled=True):
        """Per-kernel HIP events for last_launch_ms (off by default: latency path)."""
        self._check(lib().sfw_set_timing(self._h, 1 if enabled else 0), "sfw_set_timing")

    def last_launch_ms(self, which=0):
        ms = C.c_float()
        self._check(lib().sfw_last_launch_ms(self._h, which, C.byref(ms)), "sfw_last_launch_ms")
        return ms.value

    def sustained_clock_ghz(self):
        """Shader clock the last timed launch's social-force kernel ran at (0.0: nothing sampled)."""
        v = C.c_double()
        self._check(lib().sfw_last_clock_ghz(self._h, C.byref(v)), "sfw_last_clock_ghz")
        return v.value

    def set_points_capture(self, enabled=True):
        """Small grids: the scoring launch also leaves the Trajectory points (one D2H per dump, no second rollout)."""
        self._check(lib().sfw_set_points_capture(self._h, 1 if enabled else 0), "sfw_set_points_capture")

    def grid_points_batch(self, first, count, n_steps):
        """Trajectory points of `count` consecutive samples: (points[count, n_steps, 3], n_points[count])."""
        pts = np.zeros((count, n_steps, 3), dtype=np.float64)
        n = np.zeros(count, dtype=np.int32)
        self._check(lib().sfw_grid_points_batch(self._h, first, count, pts.ctypes.data, n.ctypes.data),
                    "sfw_grid_points_batch")
        return pts, n

    def grid_points(self, index, points_cap=4096):
        pts = np.zeros((points_cap, 3), dtype=np.float64)
        n = C.c_int32()
        self._check(lib().sfw_grid_points(self._h, index, pts.ctypes.data, points_cap, C.byref(n)),
                    "sfw_grid_points")
        return pts[: min(n.value, points_cap)].copy()


def plan_row_blocks(linvels, angvels, robot_state, goal_args, sim_time, num_steps, n_agents, n_ranks):
    """sfw_plan_row_blocks (host only, no device): row offsets [R + 1] of the contiguous blocks of equal planned work."""
    lin, ang = _f64(linvels), _f64(angvels)
    row0 = np.zeros(n_ranks + 1, dtype=np.int32)
    rc = lib().sfw_plan_row_blocks(lin.ctypes.data, len(lin), ang.ctypes.data, len(ang), robot_state[3], robot_state[5],
                                   goal_args[0], goal_args[2], sim_time, num_steps, n_agents, n_ranks, row0.ctypes.data)
    if rc != SFW_OK:
        raise SfwError(rc, "sfw_plan_row_blocks")
    return row0


def planned_share(linvels, angvels, robot_state, goal_args, sim_time, num_steps, n_agents):
    """Share of the algorithmic sample-steps the plan of this grid would integrate (sfw_plan_shared_prefix, host only)."""
    lin, ang = _f64(linvels), _f64(angvels)
    ends, cls, n = np.zeros(64, dtype=np.int32), np.zeros(64, dtype=np.int64), C.c_int32()
    rc = lib().sfw_plan_shared_prefix(lin.ctypes.data, len(lin), ang.ctypes.data, len(ang), robot_state[3], robot_state[5],
                                      goal_args[0], goal_args[2], sim_time, num_steps, n_agents, ends.ctypes.data,
                                      cls.ctypes.data, 64, C.byref(n))
    if rc != SFW_OK:
        raise SfwError(rc, "sfw_plan_shared_prefix")
    k = min(n.value, 64)
    if k == 0:
        return 1.0
    total = len(lin) * len(ang) * num_steps
    prev, class_steps = 0, 0
    for l in range(k):
        class_steps += int(cls[l]) * int(ends[l] - prev)
        prev = int(ends[l])
    return (class_steps + len(lin) * len(ang) * (num_steps - prev)) / total


def plan_info_of_rank(multi, r):
    """sfw_grid_plan_info of rank r's handle of a MultiScorer (after a score_grid)."""
    info = SfwPlanInfo()
    h = lib().sfw_multi_rank_handle(multi._m, r)
    rc = lib().sfw_grid_plan_info(h, C.byref(info))
    if rc != SFW_OK:
        raise SfwError(rc, "sfw_grid_plan_info")
    return info.as_dict()


class PreparedGrid:
    """See HipScorer.prepared."""

    def __init__(self, scorer, robot_state, linvels, angvels, goal_args, index_base, zero_copy=False):
        self.scorer = scorer
        self.lin, self.ang = _f64(linvels).copy(), _f64(angvels).copy()
        self.rs, self.ga = SfwRobotState(*robot_state), SfwGoalArgs(*goal_args)
        self.costs = np.empty(len(self.lin) * len(self.ang), dtype=np.float64)
        self.best, self.key = SfwBest(), SfwBestKey()
        L = lib()
        self._stage, self._launch, self._fetch = L.sfw_grid_stage, L.sfw_grid_launch, L.sfw_grid_fetch
        self._stage_args = (scorer._h, C.byref(self.rs), self.lin.ctypes.data, len(self.lin), self.ang.ctypes.data, len(self.ang),
                            C.byref(self.ga), index_base)
        self._fetch_args = (scorer._h, self.costs.ctypes.data, C.byref(self.best), C.byref(self.key))
        self._fetch_args_nocost = (scorer._h, None, C.byref(self.best), C.byref(self.key))
        scorer._grid = (len(self.lin), len(self.ang))
        self.zero_copy = zero_copy
        self._view, self._view_ptr, self._view_arr = L.sfw_grid_costs_view, None, None

    def step(self, want_costs=True):
        """Returns (costs — the SAME array every call, or with zero_copy a read-only view of the handle's own —, best, key)."""
        s = self.scorer
        rc = self._stage(*self._stage_args)
        if rc != SFW_OK:
            s._check(rc, "sfw_grid_stage")
        rc = self._launch(s._h)
        if rc != SFW_OK:
            s._check(rc, "sfw_grid_launch")
        if want_costs and self.zero_copy:
            # the vector where the launch's selection kernels left it on the host (sfw_grid_costs_view): no memcpy of it
            rc = self._fetch(*self._fetch_args_nocost)
            if rc != SFW_OK:
                s._check(rc, "sfw_grid_fetch")
            p = self._view(s._h)
            if p:
                if p != self._view_ptr:
                    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.costs.size,))
                    a.flags.writeable = False
                    self._view_ptr, self._view_arr = p, a
                return self._view_arr, self.best.as_dict(), self.key.as_tuple()
            self.zero_copy = False  # (a grid too large for the mirror: copy from here on)
        rc = self._fetch(*(self._fetch_args if want_costs else self._fetch_args_nocost))
        if rc != SFW_OK:
            s._check(rc, "sfw_grid_fetch")
        return (self.costs if want_costs else None), self.best.as_dict(), self.key.as_tuple()

    def relaunch(self):
        """launch + selection fetch of the grid staged last (no stage, no cost vector)."""
        s = self.scorer
        s._check(self._launch(s._h), "sfw_grid_launch")
        s._check(self._fetch(*self._fetch_args_nocost), "sfw_grid_fetch")
        return None, self.best.as_dict(), self.key.as_tuple()


class MultiScorer:
    """sfw_multi_*: one process driving one handle per listed device (rows of the grid split over them, one
    RCCL all-reduce(min) of the [R,5] key table).  exchange = SFW_MULTI_HOST_REDUCE lets a device be listed
    more than once (tests on a one-GPU box)."""

    def __init__(self, params=None, devices=(0,), exchange=0):
        self.params = params if params is not None else default_params()
        self._m = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = lib().sfw_multi_create(C.byref(self.params), devs, len(devices), exchange, C.byref(self._m))
        if rc != SFW_OK:
            raise SfwError(rc, "sfw_multi_create", (lib().sfw_multi_last_error(None) or b"").decode())
        self.n_ranks = lib().sfw_multi_ranks(self._m)

    def close(self):
        if getattr(self, "_m", None):
            lib().sfw_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != SFW_OK:
            raise SfwError(rc, what, (lib().sfw_multi_last_error(self._m) or b"").decode())

    def load_scene(self, scene):
        cells = np.ascontiguousarray(scene.cells, dtype=np.uint8)
        sy, sx = cells.shape
        self._check(lib().sfw_multi_set_costmap(self._m, cells.ctypes.data, sx, sy, scene.origin_x, scene.origin_y,
                                                scene.resolution), "sfw_multi_set_costmap")
        xy = _f64(scene.footprint).reshape(-1, 2)
        self._check(lib().sfw_multi_set_footprint(self._m, xy.ctypes.data if len(xy) else None, len(xy)),
                    "sfw_multi_set_footprint")
        obs = _f64(scene.obstacles if scene.obstacles is not None else np.zeros((0, 2))).reshape(-1, 2)
        n = len(scene.agents)
        self._check(lib().sfw_multi_set_agents(self._m, C.addressof(scene.agents) if n else None, n,
                                               obs.ctypes.data if len(obs) else None, len(obs)), "sfw_multi_set_agents")

    def set_params(self, params):
        self._check(lib().sfw_multi_set_params(self._m, C.byref(params)), "sfw_multi_set_params")
        self.params = params

    def score_grid(self, robot_state, linvels, angvels, goal_args, want_costs=True):
        lin, ang = _f64(linvels), _f64(angvels)
        rs, ga = SfwRobotState(*robot_state), SfwGoalArgs(*goal_args)
        costs = np.empty(len(lin) * len(ang), dtype=np.float64) if want_costs else None
        best = SfwBest()
        self._check(lib().sfw_multi_score_grid(self._m, C.byref(rs), lin.ctypes.data, len(lin), ang.ctypes.data,
                                               len(ang), C.byref(ga), costs.ctypes.data if want_costs else None,
                                               C.byref(best)), "sfw_multi_score_grid")
        return costs, best.as_dict()

    def describe(self):
        """sfw_multi_describe: devices, exchange and what the RCCL communicators report about themselves."""
        class Desc(C.Structure):
            _fields_ = [("ranks", C.c_int32), ("exchange", C.c_int32), ("communicators", C.c_int32), ("comm_size", C.c_int32),
                        ("rccl_version", C.c_int32), ("devices", C.c_int32 * 64), ("comm_devices", C.c_int32 * 64),
                        ("rccl_path", C.c_char * 512), ("rccl_found", C.c_char * 64)]

        d = Desc()
        lib().sfw_multi_describe.argtypes = [C.c_void_p, C.c_void_p]
        self._check(lib().sfw_multi_describe(self._m, C.byref(d)), "sfw_multi_describe")
        n = min(d.ranks, 64)
        return {"ranks": d.ranks, "exchange": "rccl" if d.exchange == 0 else "host_reduce", "communicators": d.communicators,
                "comm_size": d.comm_size, "rccl_version": d.rccl_version, "devices": list(d.devices[:n]),
                "comm_devices": list(d.comm_devices[:n]), "rccl_path": d.rccl_path.decode(), "rccl_found": d.rccl_found.decode()}

    def rank_rows(self, r):
        """(first row, rows) of rank r in the last score_grid."""
        a, b = C.c_int32(), C.c_int32()
        self._check(lib().sfw_multi_rank_rows(self._m, r, C.byref(a), C.byref(b)), "sfw_multi_rank_rows")
        return a.value, b.value

    def last_us(self):
        out = []
        for which in range(3):
            v = C.c_double()
            self._check(lib().sfw_multi_last_us(self._m, which, C.byref(v)), "sfw_multi_last_us")
            out.append(v.value)
        return {"enqueue_us": out[0], "exchange_us": out[1], "fetch_us": out[2]}

    def grid_points(self, index, points_cap=4096):
        pts = np.zeros((points_cap, 3), dtype=np.float64)
        n = C.c_int32()
        self._check(lib().sfw_multi_grid_points(self._m, index, pts.ctypes.data, points_cap, C.byref(n)),
                    "sfw_multi_grid_points")
        return pts[: min(n.value, points_cap)].copy()
