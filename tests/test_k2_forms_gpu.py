"""Every instantiation of the social-force kernel K2 against the oracle ("next round" 1).

K2 has two organisations of a wave — register-resident agent slots (`sfw_social_kernel<R, NS, GROUPS>`, NS = 1 for
A <= 64 with floor(64/A) samples per wave, NS = 2 for 64 < A <= 128) and all pairs flattened over the lanes
(`sfw_social_kernel_flat<R, GROUPS, CAP>`) — and the launcher picks one per launch by agent and item count: a grid
of at most 4096 samples (fewer for crowds of up to 12 agents) always runs flat.  The small grids the oracle can follow therefore never reached the
register-resident family unless it was forced.  Here every scene is scored under sfw_set_k2_form(SFW_K2_REGISTER) AND
sfw_set_k2_form(SFW_K2_FLAT), the organisation that really ran is read back from sfw_grid_plan_info, and both are
held to the oracle (f64: 1e-9, f32 forces: the north-star 1e-4) — not merely to each other.

Reference being matched: sfm::SFM.computeForces / updatePosition (group, obstacle and pair forces) reached from
/root/reference/src/sfw_planner.cpp:592-594, computeSocialWork :678-705, group ids src/sensor_interface.cpp:448-449.
"""
import dataclasses
import os

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import (SFW_K2_AUTO, SFW_K2_FLAT, SFW_K2_REGISTER, SFW_ORG_FLAT,
                                                   SFW_ORG_REGISTER_1, SFW_ORG_REGISTER_2, SFW_PRECISION_F32,
                                                   SFW_PRECISION_F64, default_params)

from test_parity_gpu import RTOL_F64, RTOL_NORTH_STAR, _assert_parity, _grouped_scene
from test_parity_holes_gpu import _rest_scene

pytestmark = pytest.mark.gpu

FORMS = [pytest.param(SFW_K2_REGISTER, id="register"), pytest.param(SFW_K2_FLAT, id="flat")]
PRECISIONS = [pytest.param(SFW_PRECISION_F64, RTOL_F64, id="f64"), pytest.param(SFW_PRECISION_F32, RTOL_NORTH_STAR, id="f32")]


def _expected_org(form, A):
    if form == SFW_K2_FLAT:
        return SFW_ORG_FLAT
    return SFW_ORG_REGISTER_1 if A <= 64 else SFW_ORG_REGISTER_2


def _check(oracle_mod, hip_mod, scene, form, prec, rtol, rs=None, n_threads=16):
    rs = scene.robot_state if rs is None else rs
    w = scene.workload
    kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
    o = oracle_mod.OracleScorer(default_params(**kw))  # the oracle is always f64
    o.load_scene(scene)
    g = hip_mod.HipScorer(default_params(precision=prec, **kw))
    g.set_k2_form(form)
    g.load_scene(scene)
    g.stage(rs, scene.linvels, scene.angvels, scene.goal_args)
    org = g.plan_info()["organisation"]
    assert org == _expected_org(form, len(scene.agents)), f"organisation {org} ran, not the requested one"
    g.launch()
    gc, gb, _ = g.fetch()
    oc, ob = o.score_grid(rs, scene.linvels, scene.angvels, scene.goal_args, n_threads=n_threads)
    assert (oc >= 0).sum() >= 4, "scene has too few valid samples to say anything"
    if prec == SFW_PRECISION_F64:
        _assert_parity(oc, ob, gc, gb, rtol)
    else:
        assert np.array_equal(oc < 0, gc < 0)
        v = oc >= 0
        rel = np.abs(gc[v] - oc[v]) / np.abs(oc[v])
        assert rel.max() <= rtol, f"max rel err {rel.max():.3e}"
        if gb["index"] != ob["index"]:  # only legal when the oracle itself cannot separate the two
            assert abs(oc[gb["index"]] - ob["cost"]) <= rtol * ob["cost"]
    return gc


# (a) group forces in the GROUPS = true instantiations: A = 21 (three samples per wave in the register form),
#     A = 60 (one sample per wave), A = 100 (two slots per lane)
@pytest.mark.parametrize("prec,rtol", PRECISIONS)
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n_people", [20, 59, 99])
def test_groups(oracle_mod, hip_mod, n_people, form, prec, rtol):
    scene = _grouped_scene(n_people, 10 + n_people)
    _check(oracle_mod, hip_mod, scene, form, prec, rtol)


# (b) slot-count boundaries without groups: G = 1 with one slot per lane (A = 52, 64), two slots per lane (A = 65,
#     111, 128), several samples per wave (A = 2, 7, 21, 32)
@pytest.mark.parametrize("prec,rtol", PRECISIONS)
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n_people", [1, 6, 20, 31, 51, 63, 64, 110, 127])
def test_agent_counts(oracle_mod, hip_mod, n_people, form, prec, rtol):
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=6, nw=7, n_people=n_people, seed=700 + n_people)
    _check(oracle_mod, hip_mod, syn.make_scene(w), form, prec, rtol)


# (b') crowds beyond the register form (A > 128): every plane capacity of the flat kernel — 208 doubles (A = 151, and 208 = its
#      last agent count + the dummy slot ... A = 207), 256 (A = 208, 231), run-time (A = 301)
@pytest.mark.parametrize("n_people", [150, 206, 207, 230, 300])
def test_large_crowd_plane_capacities(oracle_mod, hip_mod, n_people):
    w = dataclasses.replace(syn.WORKLOADS["cfg4"], nv=4, nw=5, n_people=n_people, seed=900 + n_people, sim_time=0.5)
    scene = syn.make_scene(w)
    kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
    o = oracle_mod.OracleScorer(default_params(**kw))
    o.load_scene(scene)
    g = hip_mod.HipScorer(default_params(**kw))
    g.load_scene(scene)
    g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert g.plan_info()["organisation"] == SFW_ORG_FLAT
    g.launch()
    gc, gb, _ = g.fetch()
    oc, ob = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, n_threads=32)
    assert (oc >= 0).sum() >= 4, "scene has too few valid samples to say anything"
    _assert_parity(oc, ob, gc, gb, RTOL_F64)


# (c) pairs at exact relative rest (the host-evaluated angular terms, L.agent_rest, are added in both organisations)
@pytest.mark.parametrize("prec,rtol", PRECISIONS)
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n_people,robot_moving", [(20, True), (20, False), (70, True)])
def test_relative_rest(oracle_mod, hip_mod, n_people, robot_moving, form, prec, rtol):
    scene, rs = _rest_scene(n_people, 520 + n_people, robot_moving, nv=6, nw=7)
    _check(oracle_mod, hip_mod, scene, form, prec, rtol, rs=rs)


# (d) laser points: O = 64 at A = 21 (BASELINE-adjacent `cfg2_o64`), O = 7 with groups, O = 1, O = 240 with two slots, a whole
#     720-point scan (scalar-cache loads in the wave-uniform loops; on these small grids the flat form's (agent, segment)
#     task loop reads the wave's LDS copy of the points, on a GPU-filling grid — test_gpu_filling_grid_automatic_plan[o720_51] —
#     global memory; the lane-per-agent pass with hand-pipelined scalar loads where the scan is short: (99, 33), (59, 7))
@pytest.mark.parametrize("prec,rtol", PRECISIONS)
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n_people,n_obs,grouped,seed", [(20, 64, False, 0), (20, 7, True, 60), (59, 7, True, 99), (20, 1, False, 0),
                                                         (99, 240, False, 0), (20, 720, False, 0), (50, 720, False, 0), (99, 33, True, 43), (99, 33, True, 56), (0, 16, False, 0)])
def test_laser_points(oracle_mod, hip_mod, n_people, n_obs, grouped, seed, form, prec, rtol):
    if grouped:  # seeds 43 / 56: a third of the samples end in a pedestrian contact
        scene = _grouped_scene(n_people, seed, n_obstacles=n_obs)
    else:
        w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=6, nw=7, n_people=n_people, seed=800 + n_people + n_obs,
                                n_obstacles=n_obs)
        scene = syn.make_scene(w)
    if n_people == 0 and form == SFW_K2_FLAT:
        pytest.skip("A = 1 has no flat/register distinction worth two runs")
    _check(oracle_mod, hip_mod, scene, form, prec, rtol)


# (e) a GPU-filling grid (> 4096 samples): the organisation the automatic plan picks, through the shared-prefix tree
@pytest.mark.parametrize("kind", ["groups", "rest", "o64", "plain51", "o720_51"])
def test_gpu_filling_grid_automatic_plan(oracle_mod, hip_mod, kind):
    nv, nw = (72, 64) if kind in ("plain51", "o720_51") else (96, 96)  # enough register-form waves (three samples each) for sharing to pay
    rs = None
    if kind == "groups":
        scene = _grouped_scene(20, 71, nv=nv, nw=nw)
    elif kind == "rest":
        scene, rs = _rest_scene(20, 504, True, nv=nv, nw=nw)
    elif kind == "o64":
        scene = syn.make_scene(dataclasses.replace(syn.WORKLOADS["cfg2_o64"], nv=nv, nw=nw))
    elif kind == "o720_51":  # the north-star crowd with a whole scan on a GPU-filling grid: the flat form's task loop reading
        # the points from global memory (small grids keep an LDS copy: test_laser_points), against the oracle and the register form
        scene = syn.make_scene(dataclasses.replace(syn.WORKLOADS["target_o720"], nv=nv, nw=nw))
    else:
        scene = syn.make_scene(dataclasses.replace(syn.WORKLOADS["cfg2"], nv=nv, nw=nw, n_people=50, seed=752))
    rs = scene.robot_state if rs is None else rs
    p = default_params()
    g = hip_mod.HipScorer(p)
    g.load_scene(scene)
    g.stage(rs, scene.linvels, scene.angvels, scene.goal_args)
    info = g.plan_info()
    assert info["levels"] > 0, "no shared-prefix tree on a GPU-filling grid"
    # 21 agents: three samples per register-form wave; 51 agents (the north-star crowd): the flat form wins (80 % of the lanes otherwise)
    assert info["organisation"] == (SFW_ORG_FLAT if kind in ("plain51", "o720_51") else SFW_ORG_REGISTER_1)
    g.launch()
    gc, gb, _ = g.fetch()
    o = oracle_mod.OracleScorer(p)
    o.load_scene(scene)
    rows = [0, 35, nv - 1]
    oc, _ = o.score_grid(rs, scene.linvels[rows], scene.angvels, scene.goal_args, n_threads=64)
    sub = gc.reshape(nv, nw)[rows].ravel()
    assert np.array_equal(oc < 0, sub < 0)
    v = oc >= 0
    assert v.sum() > 20
    assert np.max(np.abs(sub[v] - oc[v]) / np.abs(oc[v])) <= RTOL_F64
    # ... and the other organisation, forced, gives bit-identical costs on the whole grid
    g2 = hip_mod.HipScorer(p)
    g2.set_k2_form(SFW_K2_REGISTER if info["organisation"] == SFW_ORG_FLAT else SFW_K2_FLAT)
    g2.load_scene(scene)
    c2, b2 = g2.score_grid(rs, scene.linvels, scene.angvels, scene.goal_args)
    assert g2.plan_info()["organisation"] != info["organisation"]
    assert np.array_equal(gc, c2) and b2 == gb


def test_forms_are_bit_identical_and_the_knob_validates(hip_mod):
    """The two organisations sum every accumulator in the same order: equal bits, not just equal to 1e-9."""
    from social_force_window_planner_amd.planner import SfwError

    for n_people, n_obs in [(5, 0), (20, 0), (20, 9), (63, 0), (100, 0), (127, 3)]:
        w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=5, nw=7, n_people=n_people, seed=400 + n_people, n_obstacles=n_obs)
        scene = syn.make_scene(w)
        res = []
        for form in (SFW_K2_AUTO, SFW_K2_REGISTER, SFW_K2_FLAT):
            g = hip_mod.HipScorer(default_params())
            g.set_k2_form(form)
            g.load_scene(scene)
            res.append(g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args))
        for c, b in res[1:]:
            assert np.array_equal(c, res[0][0]) and b == res[0][1]
    g = hip_mod.HipScorer(default_params())
    with pytest.raises(SfwError):
        g.set_k2_form(7)
    # more than 128 agents: the register form does not exist, the request falls back to the flat form
    w = dataclasses.replace(syn.WORKLOADS["cfg4"], nv=2, nw=3, sim_time=0.1)
    scene = syn.make_scene(w)
    g.set_k2_form(SFW_K2_REGISTER)
    g.load_scene(scene)
    g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert g.plan_info()["organisation"] == SFW_ORG_FLAT


# (g) random crowd x scan sizes: the laser-point pass picks its form per launch (tasks over all lanes or one lane per agent,
#     LDS copy or global loads) from A, O and the item count — whatever it picks, both organisations agree bit for bit and
#     meet the oracle; small grids (lone waves) and a GPU-filling one
@pytest.mark.parametrize("seed", range(12))
def test_laser_points_random_sizes(oracle_mod, hip_mod, seed):
    rng = np.random.default_rng(4400 + seed)
    n_people = int(rng.choice([0, 1, 3, 4, 5, 11, 15, 16, 17, 31, 47, 48, 63, 64, 80, 127]))
    n_obs = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 100, 129, 255, 400, 721]))
    big = seed % 4 == 3
    nv, nw = (48, 64) if big else (int(rng.integers(1, 6)), int(rng.integers(2, 8)))
    if big:
        n_people = min(n_people, 31)  # (keeps the oracle's share of the test short)
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=nv, nw=nw, n_people=n_people, seed=4500 + seed, n_obstacles=n_obs,
                            sim_time=float(rng.choice([0.25, 0.5, 1.0])))
    scene = syn.make_scene(w)
    # a scan with structure: the ring of make_scene, every other point pulled in or pushed out
    scene.obstacles[::2] *= rng.uniform(0.5, 1.5)
    kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
    res = []
    for form in (SFW_K2_REGISTER, SFW_K2_FLAT):  # (a robot alone among the points too: 32 samples per register-form wave)
        g = hip_mod.HipScorer(default_params(**kw))
        g.set_k2_form(form)
        g.load_scene(scene)
        res.append(g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args))
    for c, b in res[1:]:
        assert np.array_equal(c, res[0][0]) and b == res[0][1], (n_people, n_obs, nv, nw)
    o = oracle_mod.OracleScorer(default_params(**kw))
    o.load_scene(scene)
    rows = np.unique(np.linspace(0, nv - 1, min(nv, 3)).round().astype(int))
    oc, _ = o.score_grid(scene.robot_state, scene.linvels[rows], scene.angvels, scene.goal_args, n_threads=os.cpu_count())
    gc = res[0][0].reshape(nv, nw)[rows].ravel()
    assert np.array_equal(oc < 0, gc < 0) and np.array_equal(oc[oc < 0], gc[gc < 0])
    v = oc >= 0
    if v.any():
        assert np.max(np.abs(gc[v] - oc[v]) / np.abs(oc[v])) <= RTOL_F64, (n_people, n_obs)

# (h) the flat form's task loop at the sizes where its two wave-uniform loops change shape (obstacle_segment_multi_uniform:
#     `rest` iterations with every lane that has points, ceil(O/16) - rest more with the lanes of the full segments; empty
#     segments for O < 16; one to four agents per lane): bit-identical with the register form, whose lanes walk the sixteen
#     segments with scalar loops — in a GPU-filling launch (points from global memory, hand-placed loads) and in a small one
#     (LDS copy of the points).  The register form meets the oracle at these sizes in (e) and (g).
@pytest.mark.parametrize("n_obs", [1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 95, 97, 250])
@pytest.mark.parametrize("filling", [False, True])
def test_task_loop_edge_sizes(hip_mod, n_obs, filling):
    nv, nw = (48, 64) if filling else (3, 5)
    for n_people in (2, 20, 37):  # 1 / 2 + 4 / 2 + 4 + 4 agents per lane in the rounds of the task loop
        w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=nv, nw=nw, n_people=n_people, seed=4700 + n_obs, n_obstacles=n_obs,
                                sim_time=0.25)
        scene = syn.make_scene(w)
        scene.obstacles[1::3] *= 0.7
        kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
        res = []
        for form in (SFW_K2_REGISTER, SFW_K2_FLAT):
            g = hip_mod.HipScorer(default_params(**kw))
            g.set_k2_form(form)
            g.load_scene(scene)
            res.append(g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args))
            if form == SFW_K2_FLAT:
                assert g.plan_info()["organisation"] == SFW_ORG_FLAT
        assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1], (n_people, n_obs, filling)
        assert (res[0][0] >= 0).any()


# (i) a register-form launch whose waves do not divide evenly over the SIMDs hands its last samples to flat-form waves on a
#     second stream (sfw_launch_social, split_point): BASELINE cfg2's 16 384 samples = 5 x 1024 waves of three + 1024 flat
#     waves.  What the plan says, and that neither the split nor the device shape it is computed for shows in any cost.
def test_split_launch_between_the_organisations(hip_mod, monkeypatch):
    monkeypatch.delenv("SFW_DEVICE_CUS", raising=False)
    w = syn.WORKLOADS["cfg2"]
    scene = syn.make_scene(w)
    kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
    res = {}
    for form in (SFW_K2_AUTO, SFW_K2_REGISTER, SFW_K2_FLAT):
        g = hip_mod.HipScorer(default_params(**kw))
        g.set_k2_form(form)
        g.load_scene(scene)
        res[form] = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        info = g.plan_info()
        if form == SFW_K2_AUTO:
            auto_info = info
        else:
            assert info["flat_samples"] == 0, info  # a forced organisation is never split
    assert auto_info["organisation"] == SFW_ORG_REGISTER_1
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:  # a whole MI355X: 1024 SIMDs
        assert auto_info["flat_samples"] == 1024, auto_info
    for form in (SFW_K2_REGISTER, SFW_K2_FLAT):
        assert np.array_equal(res[SFW_K2_AUTO][0], res[form][0]) and res[SFW_K2_AUTO][1] == res[form][1]
    # a pretended 32-CU device (128 SIMDs): 20 x 101 = 2020 samples of 21 agents = 5 x 128 x 3 + 100
    monkeypatch.setenv("SFW_DEVICE_CUS", "32")
    w2 = dataclasses.replace(w, nv=20, nw=101)
    scene2 = syn.make_scene(w2)
    out = {}
    for form in (SFW_K2_AUTO, SFW_K2_REGISTER, SFW_K2_FLAT):
        g = hip_mod.HipScorer(default_params(**kw))
        g.set_k2_form(form)
        g.load_scene(scene2)
        out[form] = g.score_grid(scene2.robot_state, scene2.linvels, scene2.angvels, scene2.goal_args)
        if form == SFW_K2_AUTO:
            info = g.plan_info()
            assert info["organisation"] == SFW_ORG_REGISTER_1 and info["flat_samples"] == 100, info
    for form in (SFW_K2_REGISTER, SFW_K2_FLAT):
        assert np.array_equal(out[SFW_K2_AUTO][0], out[form][0]) and out[SFW_K2_AUTO][1] == out[form][1]
