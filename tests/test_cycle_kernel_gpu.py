"""One launch per control cycle (csrc/sfw_kernels.hip sfw_cycle_kernel): for a small grid — the reference's own 5 x 9 samples
(ref src/sfw_planner.cpp:64-85) — pose rollout, footprint checks, pedestrian simulation and selection run as ONE kernel; the
three-kernel path (SFW_CYCLE_FUSED=0: K1 small + K2 flat + K3) is its checker: costs, sentinels, selection, Trajectory points,
point counts bit for bit, whatever the crowd, the scan, the groups, the precision mode, the step count; plus the oracle."""
import dataclasses

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import (SFW_PRECISION_F32, SFW_PRECISION_F64_STRICT, default_params)

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64), np.asarray(b, dtype=np.float64).view(np.uint64))


def _params(w, **kw):
    return default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity, **kw)


def _run(hip_mod, monkeypatch, scene, params, fused, capture=False, calls=1, goal_args=None, robot_state=None):
    monkeypatch.setenv("SFW_CYCLE_FUSED", "1" if fused else "0")
    g = hip_mod.HipScorer(params)
    g.load_scene(scene)
    if capture:
        g.set_points_capture(True)
    w = scene.workload
    out = None
    for _ in range(calls):   # (the "blocks done" counter must be back at zero after every launch)
        costs, best = g.score_grid(robot_state or scene.robot_state, scene.linvels, scene.angvels, goal_args or scene.goal_args)
        info = g.plan_info()
        pts, n = g.grid_points_batch(0, len(costs), w.n_steps)
        cur = (costs, best, pts, n, info)
        if out is not None:
            assert _same(out[0], cur[0]) and out[1] == cur[1]
        out = cur
    return out


def _check(hip_mod, monkeypatch, scene, params, capture=False, **kw):
    a = _run(hip_mod, monkeypatch, scene, params, True, capture, calls=3, **kw)
    b = _run(hip_mod, monkeypatch, scene, params, False, capture, **kw)
    assert a[4]["one_launch"] == 1 and b[4]["one_launch"] == 0
    assert _same(a[0], b[0]), np.flatnonzero(a[0] != b[0])
    assert a[1] == b[1]
    assert np.array_equal(a[3], b[3]) and _same(a[2], b[2])
    return a


@pytest.mark.parametrize("n_people", [0, 1, 5, 8, 20, 50, 62])
@pytest.mark.parametrize("n_obs", [0, 16, 60, 240])
def test_cycle_kernel_equals_the_three_kernel_path(hip_mod, monkeypatch, n_people, n_obs):
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=n_people, n_obstacles=n_obs, seed=60 + n_people)
    scene = syn.make_scene(w)
    costs, best, _, n, _ = _check(hip_mod, monkeypatch, scene, _params(w))
    assert costs[0] == -2.0 and (costs >= 0).sum() > 0  # the never-scored (0,0) sample; something survives


@pytest.mark.parametrize("steps,gran", [(1, 0.025), (6, 0.25), (7, 0.05), (40, 0.025), (65, 0.025), (200, 0.01), (512, 0.005)])
def test_step_counts(hip_mod, monkeypatch, steps, gran):
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=6, n_obstacles=30, sim_time=steps * gran, sim_granularity=gran, seed=3)
    assert w.n_steps == steps
    _check(hip_mod, monkeypatch, syn.make_scene(w), _params(w))


@pytest.mark.parametrize("precision", [SFW_PRECISION_F32, SFW_PRECISION_F64_STRICT])
def test_precision_modes(hip_mod, monkeypatch, precision):
    for n_obs in (0, 40):
        w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=12, n_obstacles=n_obs, seed=11)
        _check(hip_mod, monkeypatch, syn.make_scene(w), _params(w, precision=precision))


def test_groups_and_a_person_with_the_robots_id(hip_mod, monkeypatch):
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=14, n_obstacles=24, seed=21)
    scene = syn.make_scene(w)
    for i in range(1, 7):
        scene.agents[i].group_id = 1 + (i % 2)
    scene.agents[9].group_id = 7      # a group of one
    scene.agents[3].id = scene.agents[0].id
    a = _check(hip_mod, monkeypatch, scene, _params(w))
    for i in range(1, 7):
        scene.agents[i].group_id = -1
    scene.agents[9].group_id = -1
    b = _check(hip_mod, monkeypatch, scene, _params(w))
    assert not _same(a[0], b[0])      # the groups matter


def test_contacts_illegal_poses_and_captured_points(hip_mod, monkeypatch):
    """A crowd from 0.6 m (contacts inside the horizon), a map with discs next to the robot (illegal poses), the capturing
    launch (sfw_set_points_capture: the launch itself leaves the points; a sample the costmap rejects is integrated all the
    same so that a contact in front of the illegal pose ends its Trajectory, ref :613-627)."""
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=30, people_r_in=0.6, n_discs=20, map_size=160, seed=5)
    scene = syn.make_scene(w)
    ix = int((0.75 - scene.origin_x) / scene.resolution)
    scene.cells[:, ix:ix + 2] = 254   # a lethal wall 0.75 m ahead: the fast straight samples' footprints reach it
    for capture in (False, True):
        costs, best, pts, n, _ = _check(hip_mod, monkeypatch, scene, _params(w), capture=capture)
        assert (costs == -1.0).sum() >= 5 and (n < w.n_steps).sum() >= 5
    # a grid of 21 x 21 samples (BASELINE config 1 has no pedestrians; this one has) and 32 x 32 = the largest one launch takes
    for nv, nw in ((21, 21), (32, 32)):
        w2 = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=nv, nw=nw, n_people=9, n_obstacles=12, seed=8)
        _check(hip_mod, monkeypatch, syn.make_scene(w2), _params(w2))
    w3 = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=33, nw=32, n_people=9)   # 1056 samples: three kernels
    assert _run(hip_mod, monkeypatch, syn.make_scene(w3), _params(w3), True)[4]["one_launch"] == 0


def test_cfg1_and_the_oracle(oracle_mod, hip_mod, monkeypatch):
    """BASELINE config 1 (21 x 21 samples, no pedestrians, empty map) and the reference's 5 x 9 grid with 5 people against the
    oracle, through the one-launch path."""
    for name in ("cfg1", "ref5x9"):
        w = syn.WORKLOADS[name]
        scene = syn.make_scene(w)
        gc, gb, _, _, info = _check(hip_mod, monkeypatch, scene, _params(w))
        o = oracle_mod.OracleScorer(_params(w))
        o.load_scene(scene)
        oc, ob = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        v = oc >= 0
        assert np.array_equal(oc < 0, gc < 0) and np.array_equal(oc[~v], gc[~v])
        assert np.max(np.abs(gc[v] - oc[v]) / np.abs(oc[v])) <= 1e-9
        assert gb["index"] == ob["index"] and gb["vx"] == ob["vx"] and gb["vtheta"] == ob["vtheta"] and gb["n_valid"] == ob["n_valid"]


def test_nothing_selectable_and_a_standing_robot(hip_mod, monkeypatch):
    """Every sample rejected (a lethal map): the record says so; a robot that stands still next to a person that can never
    move takes the pinned-rest pass inside the one launch too."""
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=4, seed=9)
    scene = syn.make_scene(w)
    scene.cells[:] = 254
    costs, best, _, _, _ = _check(hip_mod, monkeypatch, scene, _params(w))
    assert best["index"] == -1 and best["n_valid"] == 0 and np.all(costs < 0)
    scene = syn.make_scene(w)
    scene.agents[2].desired_velocity = 0.0
    scene.agents[2].vx = scene.agents[2].vy = 0.0
    rs = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    scene.agents[0].vx = scene.agents[0].vy = 0.0
    _check(hip_mod, monkeypatch, scene, _params(w), robot_state=rs)
