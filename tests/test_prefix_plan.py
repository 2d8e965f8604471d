"""Host-side planning of the shared-prefix rollout (sfw_plan_shared_prefix: no handle, no device):
the class counts against an independent Python restatement of the reference's velocity recurrence
(sfw_planner.hpp:457-463), and the shape of the chosen levels."""
import ctypes as C

import numpy as np
import pytest

from social_force_window_planner_amd import planner
from social_force_window_planner_amd import synthetic as syn


def _lib():
    L = planner.lib()
    L.sfw_plan_shared_prefix.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double,
                                         C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.POINTER(C.c_int32)]
    L.sfw_plan_shared_prefix.restype = C.c_int
    return L


def _plan(lin, ang, vx0, vth0, acc_x, acc_th, sim_time, S, A, cap=64):
    lin, ang = np.ascontiguousarray(lin, dtype=np.float64), np.ascontiguousarray(ang, dtype=np.float64)
    ends = np.zeros(cap, dtype=np.int32)
    classes = np.zeros(cap, dtype=np.int64)
    n = C.c_int32(-1)
    rc = _lib().sfw_plan_shared_prefix(lin.ctypes.data, len(lin), ang.ctypes.data, len(ang), vx0, vth0, acc_x, acc_th,
                                       sim_time, S, A, ends.ctypes.data, classes.ctypes.data, cap, C.byref(n))
    assert rc == 0
    return [int(v) for v in ends[:n.value]], [int(v) for v in classes[:n.value]]


def _new_velocity(vg, vi, a_max, dt):  # reference sfw_planner.hpp:457-463
    if (vg - vi) >= 0:
        return min(vg, vi + a_max * dt)
    return max(vg, vi - a_max * dt)


def _n_classes(targets, v0, a_max, dt, p):
    """Distinct sequences of the first p velocities (bit patterns)."""
    seqs = set()
    for vg in targets:
        v, seq = v0, []
        for _ in range(p):
            v = _new_velocity(float(vg), v, a_max, dt)
            seq.append(np.float64(v).tobytes())
        seqs.add(tuple(seq))
    return len(seqs)


@pytest.mark.parametrize("name", ["cfg2", "target"])
def test_class_counts_match_the_recurrence(name):
    w = syn.WORKLOADS[name]
    scene = syn.make_scene(w)
    vx0, vth0 = scene.robot_state[3], scene.robot_state[5]
    acc_x, _, acc_th = scene.goal_args[0], scene.goal_args[1], scene.goal_args[2]
    S = w.n_steps
    ends, classes = _plan(scene.linvels, scene.angvels, vx0, vth0, acc_x, acc_th, w.sim_time, S, w.n_people + 1)
    assert len(ends) >= 1 and ends == sorted(set(ends)) and 1 <= ends[0] and ends[-1] < S
    assert classes == sorted(classes) and classes[-1] <= w.nv * w.nw
    dt = w.sim_time / S
    for p, n in zip(ends, classes):
        assert n == _n_classes(scene.linvels, vx0, acc_x, dt, p) * _n_classes(scene.angvels, vth0, acc_th, dt, p)
    # sharing has to remove a real part of the step x samples product
    shared = sum(n * (p - q) for n, p, q in zip(classes, ends, [0] + ends[:-1]))
    assert shared < 0.7 * w.nv * w.nw * ends[-1]


def test_small_grids_and_single_agents_do_not_share():
    lin, ang = syn.reference_sampler()
    assert _plan(lin, ang, 0.3, 0.0, 1.0, 1.0, 1.0, 40, 6) == ([], [])  # 45 samples: the GPU is not full
    lin, ang = syn.generalised_sampler(128, 128)
    assert _plan(lin, ang, 0.3, 0.0, 1.0, 1.0, 1.0, 40, 1) == ([], [])  # robot only: K2 has nothing to integrate twice
    assert _plan(lin, ang, 0.3, 0.0, 1.0, 1.0, 0.025, 1, 21) == ([], [])  # a single step


def test_degenerate_windows():
    lin, ang = syn.generalised_sampler(64, 64)
    # no acceleration at all: every sample keeps the current velocity for ever -> one class, one level to the end
    ends, classes = _plan(lin, ang, 0.3, 0.0, 0.0, 0.0, 1.0, 40, 21)
    assert ends == [39] and classes == [1]
    # unlimited acceleration: every sample reaches its own target in the first step -> nothing to share
    assert _plan(lin, ang, 0.3, 0.0, 1e6, 1e6, 1.0, 40, 21) == ([], [])
    # duplicate targets never separate
    lin2 = np.concatenate([lin, lin])
    ends, classes = _plan(lin2, ang, 0.3, 0.0, 1e6, 1e6, 1.0, 40, 21)
    assert ends == [39] and classes == [64 * 64]
    # long horizons: levels end within the first 48 steps
    ends, _ = _plan(lin, ang, 0.3, 0.0, 0.05, 0.05, 10.0, 400, 21)
    assert ends and ends[-1] <= 48


def test_argument_errors():
    n = C.c_int32(0)
    lin, ang = syn.generalised_sampler(64, 64)
    L = _lib()
    assert L.sfw_plan_shared_prefix(None, 64, ang.ctypes.data, 64, 0.0, 0.0, 1.0, 1.0, 1.0, 40, 5, None, None, 0, C.byref(n)) < 0
    assert L.sfw_plan_shared_prefix(lin.ctypes.data, 64, ang.ctypes.data, 64, 0.0, 0.0, 1.0, 1.0, 1.0, 0, 5, None, None, 0, C.byref(n)) < 0
    # cap = 0: only the number of levels is reported
    assert L.sfw_plan_shared_prefix(lin.ctypes.data, 64, ang.ctypes.data, 64, 0.3, 0.0, 1.0, 1.0, 1.0, 40, 21, None, None, 0, C.byref(n)) == 0
    assert n.value >= 1


def test_row_blocks_of_equal_planned_work():
    """sfw_plan_row_blocks (host only): contiguous, covering, and closer in planned work than equal row counts — BASELINE cfg5
    over 8 ranks integrates 64..76 % of its steps per equal block; the planned cut narrows that spread of WORK."""
    from social_force_window_planner_amd import planner, synthetic as syn

    w = syn.WORKLOADS["cfg5"]
    lin, ang = syn.generalised_sampler(w.nv, w.nw)
    rs, ga = (0.0, 0.0, 0.0, 0.3, 0.0, 0.0), (1.0, 0.0, 1.0, 2.0, 0.5)
    for R in (2, 4, 8):
        row0 = planner.plan_row_blocks(lin, ang, rs, ga, w.sim_time, w.n_steps, w.n_people + 1, R)
        assert row0[0] == 0 and row0[-1] == w.nv and np.all(np.diff(row0) > 0)

        def work(cuts):
            return np.array([planner.planned_share(lin[cuts[r]:cuts[r + 1]], ang, rs, ga, w.sim_time, w.n_steps, w.n_people + 1)
                             * (cuts[r + 1] - cuts[r]) for r in range(R)])

        equal = np.array([r * w.nv // R for r in range(R + 1)])
        we, wp = work(equal), work(row0)
        assert wp.max() / wp.mean() < we.max() / we.mean() or we.max() / we.mean() < 1.005
        assert wp.max() / wp.mean() < 1.02, (row0, wp)
    # nothing to share (one agent; a small grid): equal row counts
    assert list(planner.plan_row_blocks(lin[:64], ang[:16], rs, ga, 1.0, 40, 51, 4)) == [0, 16, 32, 48, 64]
    assert list(planner.plan_row_blocks(lin, ang, rs, ga, 1.0, 40, 1, 4)) == [0, 1024, 2048, 3072, 4096]
    assert list(planner.plan_row_blocks(lin[:3], ang, rs, ga, 1.0, 40, 51, 8)) == [0, 0, 0, 1, 1, 1, 2, 2, 3]


def test_the_plan_follows_the_device_shape(monkeypatch):
    """the cost model behind the shared-prefix levels counts the waves the DEVICE holds at once (4 SIMDs per
    compute unit x 5.5 waves), not those of a 256-CU MI355X: a 32-CU partition (SFW_DEVICE_CUS=32 stands in for
    hipDeviceProp_t.multiProcessorCount here — host-only planning has no device to ask) fills up with an eighth of the
    items, so coarse levels stop being 'free' there and the plan changes."""
    w = syn.WORKLOADS["cfg2"]
    scene = syn.make_scene(w)
    args = (scene.linvels, scene.angvels, scene.robot_state[3], scene.robot_state[5], scene.goal_args[0], scene.goal_args[2],
            w.sim_time, w.n_steps, w.n_people + 1)
    monkeypatch.delenv("SFW_DEVICE_CUS", raising=False)
    whole = _plan(*args)
    monkeypatch.setenv("SFW_DEVICE_CUS", "256")
    assert _plan(*args) == whole
    monkeypatch.setenv("SFW_DEVICE_CUS", "32")
    part = _plan(*args)
    assert part != whole and len(part[0]) >= 1
    monkeypatch.setenv("SFW_DEVICE_CUS", "not a number")
    assert _plan(*args) == whole
