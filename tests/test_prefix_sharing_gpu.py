"""Shared-prefix rollout (csrc/sfw_device.h: sfw_cls_agent): the first P steps are simulated once
per class of samples whose robot trajectories coincide under the acceleration limits.  It must not
show in the results: costs, sentinels, selection and Trajectory point counts are BIT-identical to
the unshared rollout, for every split step, kernel organisation, chunking and precision mode."""
import dataclasses
import os

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import SFW_PRECISION_F32, default_params

pytestmark = pytest.mark.gpu


def _score(hip_mod, scene, params, prefix, budget_mb=None, goal_args=None, want_points=False):
    """One grid call on a fresh handle created under SFW_PREFIX=<prefix> (read in sfw_create)."""
    old = {k: os.environ.get(k) for k in ("SFW_PREFIX", "SFW_TABLE_BUDGET_MB")}
    try:
        os.environ["SFW_PREFIX"] = str(prefix)
        if budget_mb is not None:
            os.environ["SFW_TABLE_BUDGET_MB"] = str(budget_mb)
        g = hip_mod.HipScorer(params)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    g.load_scene(scene)
    costs, best = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, goal_args or scene.goal_args)
    if want_points:
        _, n = g.grid_points_batch(0, len(costs), scene.workload.n_steps)
        return costs, best, n
    return costs, best


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def _params(w, **kw):
    return default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity, **kw)


@pytest.mark.parametrize("name,nv,nw,people", [("cfg2", 64, 64, 20), ("target", 64, 64, 50), ("cfg5", 16, 32, 100),
                                                ("cfg2", 48, 32, 63)])
def test_bit_identical_for_every_split(hip_mod, name, nv, nw, people):
    """Register-resident (A = 21, 64) and flat (A = 51, 101) organisations; P = 1, automatic, deep."""
    w = dataclasses.replace(syn.WORKLOADS[name], nv=nv, nw=nw, n_people=people)
    scene = syn.make_scene(w)
    p = _params(w)
    ref, bref = _score(hip_mod, scene, p, 0)
    assert (ref >= 0).sum() > 0
    for prefix in (-1, 1, 7, w.n_steps - 1, "2,5,9", "1,2,3,4,5,6,7,8", "3,30"):
        c, b = _score(hip_mod, scene, p, prefix)
        assert _same(ref, c), f"SFW_PREFIX={prefix}: costs differ"
        assert b == bref


def test_matches_oracle_with_sharing_forced(oracle_mod, hip_mod):
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=12, nw=12, n_people=9, seed=31)
    scene = syn.make_scene(w)
    p = _params(w)
    o = oracle_mod.OracleScorer(_params(w))
    o.load_scene(scene)
    oc, ob = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, n_threads=16)
    for prefix in (0, 12):
        gc, gb = _score(hip_mod, scene, p, prefix)
        v = oc >= 0
        assert np.array_equal(oc < 0, gc < 0) and np.max(np.abs(gc[v] - oc[v]) / np.abs(oc[v])) <= 1e-9
        assert gb["index"] == ob["index"] and gb["n_valid"] == ob["n_valid"]


def test_contacts_inside_the_prefix_and_points(hip_mod):
    """Dense crowd from 0.8 m: many classes end in a pedestrian contact inside the shared steps; the
    members inherit the verdict and the contact step (Trajectory point counts)."""
    w = dataclasses.replace(syn.WORKLOADS["cfg4"], nv=16, nw=16, n_people=120, people_r_in=0.8, seed=4)
    scene = syn.make_scene(w)
    p = _params(w)
    ref, bref, nref = _score(hip_mod, scene, p, 0, want_points=True)
    assert (ref == -1.0).sum() > 0
    for prefix in (3, 20, "1,2,4,8,16", -1):
        c, b, n = _score(hip_mod, scene, p, prefix, want_points=True)
        assert _same(ref, c) and b == bref and np.array_equal(n, nref)


def test_zero_sample_as_class_representative(hip_mod):
    """A decelerating robot with tight acceleration limits: every sample whose targets lie below the
    reachable window shares its first steps with the never-scored (0,0) sample, which is the first
    member (= the representative) of that class; a cluttered costmap rejects part of the grid."""
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=12, n_discs=25, seed=77)
    scene = syn.make_scene(w)
    lin, ang = syn.generalised_sampler(9, 9)  # contains (0, 0)
    scene = dataclasses.replace(scene, linvels=lin, angvels=ang, robot_state=(0.0, 0.0, 0.3, 0.5, 0.0, 0.375))
    p = _params(w)
    ga = (0.2, 0.0, 0.3, 2.0, 0.5)  # 5 mm/s and 7.5 mrad/s per step
    ref, bref = _score(hip_mod, scene, p, 0, goal_args=ga)
    assert ref[0] == -2.0
    for prefix in (1, 5, 30, "2,9,17"):
        c, b = _score(hip_mod, scene, p, prefix, goal_args=ga)
        assert _same(ref, c) and b == bref


def test_chunked_launch_groups_obstacles_and_f32(hip_mod):
    """Several chunks of whole rows (1 MiB table budget), laser points and a group, both precisions."""
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=40, nw=40, n_people=12, n_obstacles=8, seed=19)
    scene = syn.make_scene(w)
    for i in (2, 3, 4):
        scene.agents[i].group_id = 7
    for prec in ({}, {"precision": SFW_PRECISION_F32}):
        p = _params(w, **prec)
        ref, bref = _score(hip_mod, scene, p, 0)
        for prefix, budget in ((6, None), (6, 1), (-1, 1), ("2,4,9", 1)):
            c, b = _score(hip_mod, scene, p, prefix, budget_mb=budget)
            assert _same(ref, c), (prec, prefix, budget)
            assert b == bref


def test_plan_info_reports_the_split(hip_mod):
    """sfw_grid_plan_info: the automatic plan shares a prefix on a GPU-filling grid under the
    reference's acceleration limits and not on a 5 x 9 control-cycle grid; classes <= samples."""
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=64, nw=64)
    scene = syn.make_scene(w)
    g = hip_mod.HipScorer(_params(w))
    g.load_scene(scene)
    g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    info = g.plan_info()
    assert info["samples"] == 4096 and info["chunks"] == 1 and info["levels"] >= 1
    assert 1 <= info["split_step"] < w.n_steps and 0 < info["classes"] < info["samples"]
    assert 0 < info["class_steps"] < info["samples"] * info["split_step"]
    small = syn.make_scene("ref5x9")
    g.load_scene(small)
    g.stage(small.robot_state, small.linvels, small.angvels, small.goal_args)
    assert g.plan_info() == {"split_step": 0, "levels": 0, "chunks": 1, "classes": 0, "class_steps": 0, "samples": 45,
                             "organisation": 3, "flat_samples": 0,  # SFW_ORG_FLAT: a control-cycle grid runs one sample per wave
                             "one_launch": 1, "rest_noise_unreproduced": 0}  # ... of ONE kernel (sfw_cycle_kernel)
    # targets far outside the window reachable in the horizon: every sample shares every step but the last
    lin = np.linspace(5.0, 6.0, 64)
    ang = np.linspace(3.0, 4.0, 64)
    g.load_scene(scene)
    g.stage(scene.robot_state, lin, ang, scene.goal_args)
    info = g.plan_info()
    assert info["split_step"] == w.n_steps - 1 and info["classes"] == 1 and info["levels"] == 1
    assert info["class_steps"] == w.n_steps - 1
    c, b = g.score_grid(scene.robot_state, lin, ang, scene.goal_args)
    assert np.all(c == c[0]) or (c < 0).any()  # one trajectory, 4096 times (or all rejected alike)


@pytest.mark.parametrize("seed", range(24))
def test_random_scenes_random_levels(hip_mod, seed):
    """The random scenes of test_random_scenes_gpu (agent counts 0..150, footprints, laser points, step
    counts, non-default parameters) under random level sets: bitwise the plain rollout."""
    import test_random_scenes_gpu as gen

    scene, p, rs, ga, lin, ang = gen._case(500 + seed)
    rng = np.random.default_rng(seed)
    S = scene.workload.n_steps
    if S < 2 or len(scene.agents) < 2:
        pytest.skip("nothing to share")
    n_lv = int(rng.integers(1, 6))
    levels = sorted(set(int(v) for v in rng.integers(1, S, size=n_lv)))
    scene = dataclasses.replace(scene, robot_state=rs, linvels=lin, angvels=ang)
    ref, bref = _score(hip_mod, scene, p, 0, goal_args=ga)
    c, b = _score(hip_mod, scene, p, ",".join(map(str, levels)), goal_args=ga)
    assert _same(ref, c), f"levels {levels}"
    assert b == bref


def test_results_do_not_depend_on_the_device_shape(hip_mod, monkeypatch):
    """a handle scales its launch heuristics — shared-prefix cost model, the item thresholds of the K2
    organisation, the XCD-contiguous block order — to its device's compute units and XCDs.  Pretending a 32-CU, one-XCD
    partition (SFW_DEVICE_CUS / SFW_DEVICE_XCDS, read by sfw_create) changes the plan and must leave every cost, sentinel and
    the selection bit-identical."""
    for name, nv, nw in (("cfg2", 64, 64), ("cfg2", 128, 128), ("target", 64, 96)):
        w = dataclasses.replace(syn.WORKLOADS[name], nv=nv, nw=nw)
        scene = syn.make_scene(w)
        p = default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
        monkeypatch.delenv("SFW_DEVICE_CUS", raising=False)
        monkeypatch.delenv("SFW_DEVICE_XCDS", raising=False)
        g = hip_mod.HipScorer(p)
        g.load_scene(scene)
        c0, b0 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        plan0 = g.plan_info()
        plans = []
        for cus, xcds in (("32", None), ("32", "3"), ("8", "1")):
            monkeypatch.setenv("SFW_DEVICE_CUS", cus)
            if xcds:
                monkeypatch.setenv("SFW_DEVICE_XCDS", xcds)
            g2 = hip_mod.HipScorer(p)
            g2.load_scene(scene)
            c2, b2 = g2.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
            assert np.array_equal(c0, c2) and b0 == b2, (name, nv, nw, cus, xcds)
            plans.append(g2.plan_info())
        assert any(pl != plan0 for pl in plans), (plan0, plans)


@pytest.mark.parametrize("name,nv,nw,steps", [("cfg2", 64, 64, None), ("target", 48, 47, None), ("cfg3", 50, 45, None),
                                               ("cfg2", 47, 53, 13), ("cfg2", 46, 46, 1), ("cfg2", 46, 47, 600)])
def test_team_rollout_equals_the_thread_rollout(hip_mod, monkeypatch, name, nv, nw, steps):
    """K1a as teams of eight lanes per sample (sfw_rollout_team_kernel: the recurrences on three lanes, one sincos step per
    lane, positions summed by two lanes — round 6) against one thread per sample (SFW_K1A_THREADS=1, rounds 1-5): costs,
    sentinels, selection, Trajectory points and point counts bit for bit; step counts that are no multiple of eight, a
    single step, 600 steps; sample counts that are no multiple of eight (the last wave's idle teams)."""
    w = dataclasses.replace(syn.WORKLOADS[name], nv=nv, nw=nw)
    if steps is not None:
        w = dataclasses.replace(w, sim_time=steps * 0.025, sim_granularity=0.025)
    scene = syn.make_scene(w)
    assert nv * nw > 2048 and w.n_steps == (steps or w.n_steps)  # the three-kernel rollout, not the small-grid kernel
    p = _params(w)
    out = {}
    for form in ("1", "0"):
        monkeypatch.setenv("SFW_K1A_THREADS", form)
        g = hip_mod.HipScorer(p)
        g.load_scene(scene)
        costs, best = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        pts, n = g.grid_points_batch(0, nv * nw, w.n_steps)   # (> 2048 samples: the dump's K1 is the same kernel)
        out[form] = (costs, best, pts, n)
    assert (out["1"][0] >= 0).sum() > 0
    assert _same(out["1"][0], out["0"][0]) and out["1"][1] == out["0"][1]
    assert np.array_equal(out["1"][3], out["0"][3]) and _same(out["1"][2], out["0"][2])
