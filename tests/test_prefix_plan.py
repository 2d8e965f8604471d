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


def _axis(targets, v0, a_max, dt, max_p, form):
    L = planner.lib()
    t = np.ascontiguousarray(targets, dtype=np.float64)
    counts = np.zeros(max_p, dtype=np.int32)
    cls = np.full((max_p, len(t)), -1, dtype=np.int32)
    nl, closed = C.c_int32(-1), C.c_int32(-1)
    L.sfw_plan_axis_classes.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    rc = L.sfw_plan_axis_classes(t.ctypes.data, len(t), v0, a_max, dt, max_p, form, counts.ctypes.data, cls.ctypes.data,
                                 C.byref(nl), C.byref(closed))
    assert rc == 0
    return counts[:nl.value].copy(), cls[:nl.value].copy(), bool(closed.value)


def _classes_by_definition(targets, v0, a_max, dt, p):
    """Partition of the targets by the bit patterns of their first p velocities (the definition, in Python)."""
    keys = []
    for vg in targets:
        v, seq = v0, []
        for _ in range(p):
            v = _new_velocity(float(vg), v, a_max, dt)
            seq.append(np.float64(v).tobytes())
        keys.append(tuple(seq))
    return keys


def _same_partition(ids, keys):
    seen = {}
    for i, k in zip(ids, keys):
        if seen.setdefault(k, i) != i:
            return False
    return len(set(ids)) == len(set(keys))


def test_the_closed_form_of_an_axis_equals_the_walk_of_every_recurrence():
    """axis_classes::build_fast (two groups that have not reached their target + one class per reached target) against the
    generic walk (form 1) and against the definition: same class counts AND the same class of every target at every level —
    sorted and unsorted targets, duplicates, targets exactly on the start velocity and on a value the ramp passes through,
    start velocities inside / below / above the window, ramps that overshoot in the first step."""
    rng = np.random.default_rng(606)
    cases = []
    for name in ("cfg2", "target", "cfg3"):
        w = syn.WORKLOADS[name]
        sc = syn.make_scene(w)
        dt = w.sim_time / w.n_steps
        cases.append((sc.linvels, sc.robot_state[3], sc.goal_args[0], dt, min(w.n_steps - 1, 48)))
        cases.append((sc.angvels, sc.robot_state[5], sc.goal_args[2], dt, min(w.n_steps - 1, 48)))
    lin, ang = syn.reference_sampler()
    cases.append((ang, 0.1, 1.0, 0.025, 39))            # the reference's alternating angular samples: not sorted
    cases.append((lin, 0.35, 1.0, 0.025, 39))           # start velocity ON a target
    cases.append((np.linspace(-1, 1, 41), -0.1, 2.0, 0.025, 30))   # -0.1 + 2 x 0.05 passes through 0.0 = a target
    cases.append((np.linspace(-1, 1, 41), 3.0, 1.0, 0.1, 48))      # start above the whole window
    cases.append((np.linspace(-1, 1, 41), -3.0, 1.0, 0.1, 48))     # ... below it
    cases.append((np.repeat(np.linspace(0, 1, 9), 3), 0.5, 1.0, 0.05, 20))  # duplicates never separate
    cases.append((np.linspace(0, 1, 33), 0.5, 1e6, 0.1, 10))       # everything reached in the first step
    for _ in range(60):
        n = int(rng.integers(1, 90))
        t = rng.uniform(-2, 2, n)
        if rng.random() < 0.5:
            t = np.round(t * 8) / 8 + 0.0                # many exact duplicates and exactly representable steps (no -0.0:
                                                         # the next test)
        v0 = float(rng.choice([rng.uniform(-2.5, 2.5), t[rng.integers(n)], 0.0, 0.125 * rng.integers(-16, 16)]))
        a = float(rng.choice([0.5, 1.0, 2.5, 0.125 / 0.025]))
        dt = float(rng.choice([0.025, 0.05, 0.25]))
        cases.append((t, v0, a, dt, int(rng.integers(1, 49))))
    n_closed = 0
    for t, v0, a, dt, max_p in cases:
        cf, clf, closed = _axis(t, v0, a, dt, max_p, 0)
        cg, clg, closed_g = _axis(t, v0, a, dt, max_p, 1)
        assert not closed_g
        n_closed += closed
        assert np.array_equal(cf, cg), (t, v0, a, dt)
        assert np.array_equal(clf, clg), (t, v0, a, dt)     # same numbering too (both number the classes in target order)
        for p in {1, min(2, len(cf)), len(cf) // 2 + 1, len(cf)}:
            keys = _classes_by_definition(t, v0, a, dt, p)
            assert cf[p - 1] == len(set(keys)) and _same_partition(clf[p - 1], keys), (p, t, v0, a, dt)
    assert n_closed >= len(cases) - 2  # the closed form is what runs


def test_inputs_outside_the_closed_forms_premises_take_the_walk():
    t = np.linspace(-1, 1, 17)
    # no acceleration: nothing moves, one class for ever
    c, cl, closed = _axis(t, 0.3, 0.0, 0.025, 10, 0)
    assert not closed and list(c) == [1] * 10
    # a step below half an ulp of the start velocity: u_p = v0 for ever
    c, cl, closed = _axis(t, 1e6, 1e-12, 0.025, 5, 0)
    assert not closed and np.array_equal(c, _axis(t, 1e6, 1e-12, 0.025, 5, 1)[0])
    # a negative zero among the targets (the walk compares bit patterns)
    t2 = t.copy()
    t2[8] = -0.0
    c, cl, closed = _axis(t2, 0.3, 1.0, 0.025, 10, 0)
    assert not closed and np.array_equal(cl, _axis(t2, 0.3, 1.0, 0.025, 10, 1)[1])
