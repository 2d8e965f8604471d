"""ROS-free core of SFWPlannerNode::transformGlobalPlan (reference
src/sfw_planner_node.cpp:119-185), checked against an independent numpy restatement."""
import ctypes as C
import math

import numpy as np
import pytest

from social_force_window_planner_amd import host


@pytest.fixture(scope="module")
def L():
    host.build()
    lib = host.lib()
    lib.sfwh_transform_global_plan.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double,
                                               C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int32,
                                               C.POINTER(C.c_int32)]
    return lib


def _call(L, plan, robot, sx, sy, res, tf):
    plan = np.ascontiguousarray(plan, dtype=np.float64).reshape(-1, 3).copy()
    robot = np.ascontiguousarray(robot, dtype=np.float64)
    out = np.zeros((max(len(plan), 1), 3))
    rem = C.c_int32()
    n = L.sfwh_transform_global_plan(plan.ctypes.data if len(plan) else None, len(plan), robot.ctypes.data, sx, sy, res,
                                     tf[0], tf[1], tf[2], out.ctypes.data, len(out), C.byref(rem))
    return n, out[:max(n, 0)], plan[:rem.value]


def _numpy(plan, robot, sx, sy, res, tf):
    plan = np.asarray(plan, dtype=np.float64).reshape(-1, 3)
    d = np.hypot(plan[:, 0] - robot[0], plan[:, 1] - robot[1])
    b = int(np.argmin(d))  # first minimum
    lim = max(sx, sy) * res / 2.0
    e = b
    while e < len(plan) and not d[e] > lim:
        e += 1
    c, s = math.cos(tf[2]), math.sin(tf[2])
    seg = plan[b:e]
    out = np.column_stack([tf[0] + c * seg[:, 0] - s * seg[:, 1], tf[1] + s * seg[:, 0] + c * seg[:, 1], seg[:, 2] + tf[2]])
    return out, plan[b:]


def test_prune_and_transform(L):
    rng = np.random.default_rng(3)
    for _ in range(50):
        n = int(rng.integers(1, 60))
        t = np.linspace(0, rng.uniform(1, 30), n)
        plan = np.column_stack([t * math.cos(0.3), t * math.sin(0.3) + 0.2 * np.sin(t), rng.uniform(-1, 1, n)])
        robot = plan[int(rng.integers(0, n)), :2] + rng.uniform(-0.3, 0.3, 2)
        tf = (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), float(rng.uniform(-1, 1)))
        sx, sy, res = int(rng.integers(40, 400)), int(rng.integers(40, 400)), 0.05
        m, out, rest = _call(L, plan, robot, sx, sy, res, tf)
        eo, er = _numpy(plan, robot, sx, sy, res, tf)
        if len(eo) == 0:
            assert m == -2  # "Resulting plan has 0 poses in it."
            continue
        assert m == len(eo) and np.allclose(out, eo, atol=1e-12)
        assert np.allclose(rest, er, atol=1e-12)  # poses behind the robot are dropped from the stored plan


def test_exceptions(L):
    assert _call(L, np.zeros((0, 3)), [0, 0], 100, 100, 0.05, (0, 0, 0))[0] == -1   # "Received plan with zero length"
    # robot farther than half the costmap from every pose: nothing to follow
    m, _, rest = _call(L, [[10, 0, 0], [11, 0, 0]], [0, 0], 100, 100, 0.05, (0, 0, 0))
    assert m == -2 and len(rest) == 2
    # ties: the FIRST closest pose starts the local plan
    m, out, rest = _call(L, [[-1, 0, 0], [1, 0, 0], [2, 0, 0]], [0, 0], 200, 200, 0.05, (0, 0, 0))
    assert m == 3 and out[0, 0] == -1.0 and len(rest) == 3
