"""Error conventions of the C ABI on a real device (include/sfw_hip.h): status
codes, last_error text, invalid trajectory = data (-1.0), call-order checks."""
import ctypes as C
import dataclasses

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import (SFW_ERR_INVALID_ARG, SFW_ERR_STATE, SFW_ERR_UNSUPPORTED, SFW_OK,
                                                   SfwAgent, SfwBest, SfwGoalArgs, SfwRobotState, default_params)

pytestmark = pytest.mark.gpu


def test_call_order_and_argument_errors(hip_mod):
    L = hip_mod.lib()
    g = hip_mod.HipScorer(default_params())
    scene = syn.make_scene("ref5x9")
    rs, ga = SfwRobotState(*scene.robot_state), SfwGoalArgs(*scene.goal_args)
    lin, ang = scene.linvels, scene.angvels
    costs = np.zeros(45)
    best = SfwBest()
    # scoring before a costmap was set
    rc = L.sfw_score_grid(g._h, C.byref(rs), lin.ctypes.data, 5, ang.ctypes.data, 9, C.byref(ga), costs.ctypes.data,
                          C.byref(best))
    assert rc == SFW_ERR_STATE and b"costmap" in L.sfw_last_error(g._h)
    assert L.sfw_grid_launch(g._h) == SFW_ERR_STATE
    assert L.sfw_grid_fetch(g._h, None, C.byref(best), None) == SFW_ERR_STATE
    g.load_scene(scene)
    assert L.sfw_score_grid(g._h, C.byref(rs), lin.ctypes.data, 0, ang.ctypes.data, 9, C.byref(ga), None, None) == SFW_ERR_INVALID_ARG
    assert L.sfw_score_grid(g._h, None, lin.ctypes.data, 5, ang.ctypes.data, 9, C.byref(ga), None, None) == SFW_ERR_INVALID_ARG
    assert L.sfw_set_costmap(g._h, None, 10, 10, 0.0, 0.0, 0.05) == SFW_ERR_INVALID_ARG
    assert L.sfw_set_costmap(g._h, scene.cells.ctypes.data, 10, 10, 0.0, 0.0, 0.0) == SFW_ERR_INVALID_ARG
    assert L.sfw_set_footprint(g._h, None, 3) == SFW_ERR_INVALID_ARG
    # outputs are optional
    assert L.sfw_score_grid(g._h, C.byref(rs), lin.ctypes.data, 5, ang.ctypes.data, 9, C.byref(ga), None, None) == SFW_OK
    # a failed set_agents leaves the previous agent set in place
    assert L.sfw_set_agents(g._h, None, 2, None, 0) == SFW_ERR_INVALID_ARG
    c1, b1 = g.score_grid(scene.robot_state, lin, ang, scene.goal_args)
    g2 = hip_mod.HipScorer(default_params())
    g2.load_scene(scene)
    c2, b2 = g2.score_grid(scene.robot_state, lin, ang, scene.goal_args)
    assert np.array_equal(c1, c2) and b1 == b2


def test_zero_agents_and_reuse_of_a_handle(oracle_mod, hip_mod):
    """A = 0 (no agent vector at all): social work is 0; then the same handle is
    reused with a different costmap size, agent count and grid size."""
    scene = syn.make_scene("cfg1")
    p = default_params(sim_time=0.5)
    g = hip_mod.HipScorer(p)
    o = oracle_mod.OracleScorer(default_params(sim_time=0.5))
    for s in (g, o):
        s.set_costmap(scene.cells, scene.origin_x, scene.origin_y, scene.resolution)
        s.set_footprint(scene.footprint)
        s.set_agents((SfwAgent * 0)())
    gc, gb = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    oc, ob = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert np.allclose(gc, oc, rtol=1e-13, atol=0) and gb["index"] == ob["index"]
    for name in ("ref5x9", "cfg2"):
        w = dataclasses.replace(syn.WORKLOADS[name], nv=min(syn.WORKLOADS[name].nv, 16), nw=9 if name == "ref5x9" else 16)
        if name == "ref5x9":
            w = syn.WORKLOADS[name]
        sc = syn.make_scene(w)
        pp = default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
        g.set_params(pp)
        o.set_params(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
        g.load_scene(sc)
        o.load_scene(sc)
        gc, gb = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
        oc, ob = o.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
        v = oc >= 0
        assert np.array_equal(oc < 0, gc < 0) and np.max(np.abs(gc[v] - oc[v]) / np.abs(oc[v])) <= 1e-9
        assert gb["index"] == ob["index"]


def test_many_agents_use_the_flat_kernel(oracle_mod, hip_mod):
    """A = 300 > 128 agents: the LDS-resident flat kernel; a few samples only."""
    w = dataclasses.replace(syn.WORKLOADS["cfg4"], nv=2, nw=3, n_people=299, seed=9)
    scene = syn.make_scene(w)
    p = default_params()
    g = hip_mod.HipScorer(p)
    g.load_scene(scene)
    o = oracle_mod.OracleScorer(default_params())
    o.load_scene(scene)
    gc, gb = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    oc, ob = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, n_threads=6)
    assert np.array_equal(oc < 0, gc < 0)
    v = oc >= 0
    if v.any():
        assert np.max(np.abs(gc[v] - oc[v]) / np.abs(oc[v])) <= 1e-9
    assert gb["index"] == ob["index"]


def test_timing_is_opt_in(hip_mod):
    """sfw_last_launch_ms needs sfw_set_timing(1) before the launch; timed and untimed launches
    give the same costs."""
    L = hip_mod.lib()
    scene = syn.make_scene("ref5x9")
    g = hip_mod.HipScorer(default_params())
    g.load_scene(scene)
    c0, b0 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    ms = C.c_float(0)
    assert L.sfw_last_launch_ms(g._h, 0, C.byref(ms)) == SFW_ERR_STATE and b"timing" in L.sfw_last_error(g._h)
    g.set_timing(True)
    c1, b1 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert np.array_equal(c0, c1) and b0 == b1
    total = g.last_launch_ms(0)
    parts = [g.last_launch_ms(k) for k in (1, 2, 3)]
    assert total > 0 and all(p >= 0 for p in parts) and sum(parts) <= total * 1.05
    assert L.sfw_last_launch_ms(g._h, 7, C.byref(ms)) == SFW_ERR_INVALID_ARG
    # the shader clock a wave of the social-force launch measured for itself (s_memtime against s_memrealtime)
    assert 1.0 < g.sustained_clock_ghz() < 2.6
    g.set_timing(False)
    g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    ghz = C.c_double()
    assert L.sfw_last_clock_ghz(g._h, C.byref(ghz)) == SFW_ERR_STATE
    # no agents: no social-force launch, nothing sampled
    g.set_timing(True)
    g.set_agents((SfwAgent * 0)())
    g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert g.sustained_clock_ghz() == 0.0


def test_world_changes_between_stage_and_launch(oracle_mod, hip_mod):
    """sfw_set_agents after a stage takes effect at the NEXT stage: a launch in between still
    scores the staged world (and does not read past the uploaded agent arrays)."""
    w5 = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=5)
    w40 = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=40, seed=8)
    s5, s40 = syn.make_scene(w5), syn.make_scene(w40)
    g = hip_mod.HipScorer(default_params())
    g.load_scene(s5)
    g.stage(s5.robot_state, s5.linvels, s5.angvels, s5.goal_args)
    g.set_agents(s40.agents, s40.obstacles)  # more agents than the device copy holds
    g.launch()
    c_staged, b_staged, _ = g.fetch(want_costs=True)
    o = oracle_mod.OracleScorer(default_params())
    o.load_scene(s5)
    oc, ob = o.score_grid(s5.robot_state, s5.linvels, s5.angvels, s5.goal_args)
    v = oc >= 0
    assert np.array_equal(oc < 0, c_staged < 0) and np.max(np.abs(c_staged[v] - oc[v]) / np.abs(oc[v])) <= 1e-9
    assert b_staged["index"] == ob["index"]
    # the next stage picks the new agent set up
    c2, b2 = g.score_grid(s5.robot_state, s5.linvels, s5.angvels, s5.goal_args)
    o.set_agents(s40.agents, s40.obstacles)
    oc2, ob2 = o.score_grid(s5.robot_state, s5.linvels, s5.angvels, s5.goal_args)
    v = oc2 >= 0
    assert np.array_equal(oc2 < 0, c2 < 0) and np.max(np.abs(c2[v] - oc2[v]) / np.abs(oc2[v])) <= 1e-9
    assert b2["index"] == ob2["index"]


def test_non_finite_inputs_are_refused(hip_mod):
    """"never NaN" (include/sfw_hip.h) holds because NaN / Inf never get in: agents, laser points, robot state, goal
    arguments and sample velocities are checked on the host (O(A + O + nv + nw)) and refused with
    SFW_ERR_INVALID_ARG; the handle keeps working with its previous world state."""
    L = hip_mod.lib()
    scene = syn.make_scene("ref5x9")
    g = hip_mod.HipScorer(default_params())
    g.load_scene(scene)
    c0, b0 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    lin, ang = scene.linvels, scene.angvels
    best = SfwBest()
    n = len(scene.agents)
    for field, bad in (("x", float("nan")), ("vy", float("inf")), ("radius", float("nan")),
                       ("desired_velocity", float("-inf")), ("goal_x", float("nan"))):
        agents = (SfwAgent * n)(*scene.agents)
        setattr(agents[2], field, bad)
        assert L.sfw_set_agents(g._h, C.addressof(agents), n, None, 0) == SFW_ERR_INVALID_ARG, field
        assert b"agent 2" in L.sfw_last_error(g._h)
    # a goal that is not there may hold anything
    agents = (SfwAgent * n)(*scene.agents)
    agents[2].has_goal, agents[2].goal_x = 0, float("nan")
    assert L.sfw_set_agents(g._h, C.addressof(agents), n, None, 0) == SFW_OK
    g.load_scene(scene)
    obs = np.array([[1.0, 2.0], [float("nan"), 0.0]])
    assert L.sfw_set_agents(g._h, C.addressof(scene.agents), n, obs.ctypes.data, 2) == SFW_ERR_INVALID_ARG
    ga = SfwGoalArgs(*scene.goal_args)
    for k in range(6):
        v = list(scene.robot_state)
        v[k] = float("nan")
        rs = SfwRobotState(*v)
        assert L.sfw_score_grid(g._h, C.byref(rs), lin.ctypes.data, 5, ang.ctypes.data, 9, C.byref(ga), None,
                                C.byref(best)) == SFW_ERR_INVALID_ARG
    rs = SfwRobotState(*scene.robot_state)
    gbad = SfwGoalArgs(*scene.goal_args)
    gbad.wpx = float("inf")
    assert L.sfw_score_grid(g._h, C.byref(rs), lin.ctypes.data, 5, ang.ctypes.data, 9, C.byref(gbad), None, None) == SFW_ERR_INVALID_ARG
    lbad = lin.copy()
    lbad[3] = float("nan")
    assert L.sfw_score_grid(g._h, C.byref(rs), lbad.ctypes.data, 5, ang.ctypes.data, 9, C.byref(ga), None, None) == SFW_ERR_INVALID_ARG
    # a NaN in a large sample vector reaches the shared-prefix planner's sort in no case
    big_l, big_a = syn.generalised_sampler(96, 96)
    big_a = big_a.copy()
    big_a[50] = float("nan")
    assert L.sfw_grid_stage(g._h, C.byref(rs), big_l.ctypes.data, 96, big_a.ctypes.data, 96, C.byref(ga), 0) == SFW_ERR_INVALID_ARG
    cost = C.c_double()
    assert L.sfw_score_one(g._h, C.byref(rs), 0.3, float("nan"), 0.1, C.byref(ga), C.byref(cost), None, 0, None) == SFW_ERR_INVALID_ARG
    c1, b1 = g.score_grid(scene.robot_state, lin, ang, scene.goal_args)
    assert np.array_equal(c0, c1) and b0 == b1 and np.all(np.isfinite(c1))


def test_unsupported_agent_sets(hip_mod):
    """What the device cannot (or, for parity, must not) take is SFW_ERR_UNSUPPORTED, not a failed launch: more agents
    than one wave's LDS holds (about 2000), more than the 16-bit plane offsets of the pair table reach.  A person that can never
    move (desired_velocity <= 0, the reference's people_velocity_ = 0) is NOT refused (tests/test_parity_holes_gpu.py scores one)."""
    from social_force_window_planner_amd.planner import SfwError

    L = hip_mod.lib()
    scene = syn.make_scene("ref5x9")
    g = hip_mod.HipScorer(default_params())
    g.load_scene(scene)
    n = len(scene.agents)
    agents = (SfwAgent * n)(*scene.agents)
    agents[3].desired_velocity = 0.0
    assert L.sfw_set_agents(g._h, C.addressof(agents), n, None, 0) == SFW_OK
    agents[3].desired_velocity = 1.0
    agents[0].desired_velocity = 0.0  # the robot is not integrated by the social-force model: anything goes
    assert L.sfw_set_agents(g._h, C.addressof(agents), n, None, 0) == SFW_OK
    for count, where in ((9000, "set_agents"), (3000, "set_agents"), (1500, "ok")):
        many = (SfwAgent * count)()
        for i in range(count):
            a = many[i]
            a.x, a.y, a.vx, a.vy = 3.0 + 0.001 * i, 2.0 + 0.0013 * (i % 97), 0.1, 1e-5 * i
            a.goal_x, a.goal_y, a.goal_radius, a.desired_velocity, a.radius, a.has_goal, a.id, a.group_id = 9.0, 9.0, 0.35, 1.0, 0.35, 1, i, -1
        rc = L.sfw_set_agents(g._h, C.addressof(many), count, None, 0)
        if where == "set_agents":   # refused before the all-pairs scan of the set and long before a launch
            assert rc == SFW_ERR_UNSUPPORTED and (b"LDS" in L.sfw_last_error(g._h) or b"8190" in L.sfw_last_error(g._h))
        else:                       # 1500 agents fit (flat form, run-time plane capacity): scored
            assert rc == SFW_OK
            c, b = g.score_grid(scene.robot_state, scene.linvels[:2], scene.angvels[:3], scene.goal_args)
            assert np.all(np.isfinite(c))
    g.load_scene(scene)
    c, b = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert (c >= 0).any()
