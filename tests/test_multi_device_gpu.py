"""Single-process multi-device entry of the C ABI (include/sfw_hip.h: sfw_multi_*) and its use by the C++ host
mirror (SFWPlanner::setDevices).  On the one-GPU box: R = 1 over RCCL (ncclCommInitAll + ncclAllReduce(min) really
run) and R = 2, 3 on handles sharing the device behind the host-side reduce, all bit-equal to sfw_score_grid."""
import dataclasses
import math
import os
import subprocess

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import SFW_MULTI_HOST_REDUCE, SFW_MULTI_RCCL, default_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(nv, nw, n_people=9, seed=71, **kw):
    return syn.make_scene(dataclasses.replace(syn.WORKLOADS["cfg2"], nv=nv, nw=nw, n_people=n_people, seed=seed, **kw))


@pytest.mark.parametrize("devices,exchange", [((0,), SFW_MULTI_RCCL), ((0, 0), SFW_MULTI_HOST_REDUCE),
                                              ((0, 0, 0), SFW_MULTI_HOST_REDUCE)])
@pytest.mark.parametrize("nv,nw", [(24, 16), (72, 72), (2, 9)])
def test_multi_equals_single(hip_mod, devices, exchange, nv, nw):
    scene = _scene(nv, nw)
    p = default_params()
    g = hip_mod.HipScorer(p)
    g.load_scene(scene)
    c1, b1 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    m = hip_mod.MultiScorer(p, devices=devices, exchange=exchange)
    assert m.n_ranks == len(devices)
    m.load_scene(scene)
    c2, b2 = m.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    assert np.array_equal(c1, c2) and b1 == b2
    # without the cost vector: the selection alone, from the exchanged table
    _, b3 = m.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, want_costs=False)
    assert b3 == b1
    us = m.last_us()
    assert us["exchange_us"] > 0 and us["enqueue_us"] > 0
    # Trajectory points route to the rank that holds the sample
    for idx in (1, nv * nw - 1, (nv // 2) * nw + 3):
        assert np.array_equal(m.grid_points(idx), g.grid_points(idx))


@pytest.mark.parametrize("nv,nw,R", [(256, 128, 4), (64, 256, 8), (130, 96, 8), (5, 9, 8)])
def test_many_ranks_share_the_column_plan(hip_mod, nv, nw, R):
    """R = 4 / 8 ranks staged and launched by the handle's worker threads, the shared-prefix classes of the column
    axis computed once for all of them (256 x 128 over 4 ranks: ~8192 samples per rank, every rank plans a prefix
    tree): costs, selection and plan-dependent results bit-equal to one sfw_score_grid; 5 rows over 8 ranks: three
    ranks hold nothing."""
    scene = _scene(nv, nw, n_people=12, seed=72)
    p = default_params()
    g = hip_mod.HipScorer(p)
    g.load_scene(scene)
    c1, b1 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    m = hip_mod.MultiScorer(p, devices=(0,) * R, exchange=SFW_MULTI_HOST_REDUCE)
    m.load_scene(scene)
    for _ in range(3):  # the workers are reused call after call
        c2, b2 = m.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        assert np.array_equal(c1, c2) and b1 == b2
    # contiguous blocks covering the grid; of equal planned work (sfw_plan_row_blocks) where the prefix tree shares steps
    blocks = [m.rank_rows(r) for r in range(R)]
    assert blocks[0][0] == 0 and sum(n for _, n in blocks) == nv
    assert all(blocks[r][0] + blocks[r][1] == blocks[r + 1][0] for r in range(R - 1))
    if (nv // R) * nw >= 4096:
        assert any(hip_mod.plan_info_of_rank(m, r)["levels"] > 0 for r in range(R))  # (a block cut below 4096 samples plans none)
        cuts = hip_mod.plan_row_blocks(scene.linvels, scene.angvels, scene.robot_state, scene.goal_args, 1.0, 40,
                                       len(scene.agents), R)
        assert [b[0] for b in blocks] == list(cuts[:-1])
    else:
        assert [b[0] for b in blocks] == [r * nv // R for r in range(R)]
    # another robot state and another column axis: the lent column classes must not outlive their call
    rs2 = (0.0, 0.0, 0.0, float(np.float32(0.1)), 0.0, float(np.float32(0.2)))
    ang2 = scene.angvels[::-1].copy()
    c3, b3 = g.score_grid(rs2, scene.linvels, ang2, scene.goal_args)
    c4, b4 = m.score_grid(rs2, scene.linvels, ang2, scene.goal_args)
    assert np.array_equal(c3, c4) and b3 == b4


def test_two_multi_planners_created_concurrently(hip_mod):
    """librccl is resolved once however many threads create their first RCCL planner at the same time."""
    import threading

    scene = _scene(12, 9)
    p = default_params()
    g = hip_mod.HipScorer(p)
    g.load_scene(scene)
    c1, b1 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    out = [None, None]

    def work(k):
        m = hip_mod.MultiScorer(p, devices=(0,), exchange=SFW_MULTI_RCCL)
        m.load_scene(scene)
        out[k] = m.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        m.close()

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c, b in out:
        assert np.array_equal(c, c1) and b == b1


def test_multi_checks_the_lds_budget_before_any_rank_launches(hip_mod):
    """An agent set that does not fit one wave's LDS is refused for the whole grid up front (SFW_ERR_UNSUPPORTED),
    not after some ranks have been launched."""
    from social_force_window_planner_amd._abi import SFW_ERR_UNSUPPORTED, SfwAgent
    from social_force_window_planner_amd.planner import SfwError

    scene = _scene(8, 9)
    n = 3000
    agents = (SfwAgent * n)()
    for i in range(n):
        a = agents[i]
        a.x, a.y, a.vx, a.vy = 3.0 + 0.01 * i, 2.0 + 0.013 * (i % 97), 0.1, 0.001 * i
        a.goal_x, a.goal_y, a.goal_radius, a.desired_velocity, a.radius, a.has_goal, a.id, a.group_id = 9.0, 9.0, 0.35, 1.0, 0.35, 1, i, -1
    scene.agents = agents
    m = hip_mod.MultiScorer(default_params(), devices=(0, 0), exchange=SFW_MULTI_HOST_REDUCE)
    with pytest.raises(SfwError) as e:   # sfw_multi_set_agents already: no rank has staged or launched anything
        m.load_scene(scene)
    assert e.value.status == SFW_ERR_UNSUPPORTED and "LDS" in str(e.value)
    g = hip_mod.HipScorer(default_params())
    with pytest.raises(SfwError) as e:
        g.load_scene(scene)
    assert e.value.status == SFW_ERR_UNSUPPORTED
    # a set that fits alone but not with its laser points in the register form's per-wave copy is refused by the grid
    # call, before any rank launches (sfw_multi_score_grid checks every rank's item count first)
    ok_scene = _scene(8, 9)
    m.load_scene(ok_scene)
    c, b = m.score_grid(ok_scene.robot_state, ok_scene.linvels, ok_scene.angvels, ok_scene.goal_args)
    assert np.all(np.isfinite(c))


def test_multi_tie_across_ranks_and_all_invalid(hip_mod):
    """A full tie between samples of different ranks must resolve as in one launch (later iterate wins), and a
    grid without a selectable sample returns index -1 with the summed n_valid."""
    scene = syn.make_scene(dataclasses.replace(syn.WORKLOADS["cfg1"], nv=8, nw=9, footprint="point", map_size=400))
    p = default_params(sim_time=0.5, distance_weight=0.0, angle_weight=0.0)  # cost depends on the row only
    lin = np.array([0.1, 0.3, 0.3, 0.5, 0.5, 0.5, 0.2, 0.5])  # equal target speeds in different ranks' blocks
    g = hip_mod.HipScorer(p)
    g.load_scene(scene)
    c1, b1 = g.score_grid(scene.robot_state, lin, scene.angvels, scene.goal_args)
    m = hip_mod.MultiScorer(p, devices=(0, 0, 0), exchange=SFW_MULTI_HOST_REDUCE)
    m.load_scene(scene)
    c2, b2 = m.score_grid(scene.robot_state, lin, scene.angvels, scene.goal_args)
    assert np.array_equal(c1, c2) and b1 == b2
    assert (c1 == c1[b1["index"]]).sum() > 9  # the winner's cost occurs in several rows
    scene.cells[:] = 254
    g.load_scene(scene)
    m.load_scene(scene)
    c1, b1 = g.score_grid(scene.robot_state, lin, scene.angvels, scene.goal_args)
    c2, b2 = m.score_grid(scene.robot_state, lin, scene.angvels, scene.goal_args)
    assert np.array_equal(c1, c2) and b1 == b2 and b2["index"] == -1 and b2["n_valid"] == 0


def test_multi_argument_errors(hip_mod):
    import ctypes as C

    from social_force_window_planner_amd._abi import SFW_ERR_INVALID_ARG

    L = hip_mod.lib()
    p = default_params()
    h = C.c_void_p()
    two = (C.c_int * 2)(0, 0)
    assert L.sfw_multi_create(C.byref(p), two, 2, SFW_MULTI_RCCL, C.byref(h)) == SFW_ERR_INVALID_ARG  # one rank per device
    assert L.sfw_multi_create(C.byref(p), two, 0, SFW_MULTI_RCCL, C.byref(h)) == SFW_ERR_INVALID_ARG
    assert L.sfw_multi_create(C.byref(p), None, 1, SFW_MULTI_RCCL, C.byref(h)) == SFW_ERR_INVALID_ARG
    bad = (C.c_int * 1)(99)
    assert L.sfw_multi_create(C.byref(p), bad, 1, SFW_MULTI_HOST_REDUCE, C.byref(h)) != 0 and not h.value
    assert L.sfw_multi_destroy(None) == 0 and L.sfw_multi_ranks(None) == 0


@pytest.mark.parametrize("devices,host_reduce", [((0,), False), ((0, 0), True)])
def test_host_planner_on_several_ranks(oracle_mod, devices, host_reduce):
    """The C++ SFWPlanner in multi-device mode drives exactly like the single-device planner and the oracle."""
    from social_force_window_planner_amd import host
    from social_force_window_planner_amd._abi import BRANCH_GRID, default_ctrl_params

    host.build()
    scene = syn.make_scene("ref5x9")
    ctrl = default_ctrl_params()
    single = host.HostPlanner(ctrl, scene)
    multi = host.HostPlanner(ctrl, scene)
    multi.set_devices(list(devices), host_reduce)
    lin, ang = syn.generalised_sampler(12, 9)
    plan = [[x, 0.15 * x, 0.2] for x in np.linspace(0.0, 4.0, 17)]
    for pl in (single, multi):
        pl.set_sample_sets(lin, ang)
        pl.update_plan(plan)
    pose, vel = np.array([0.0, 0.0, 0.0]), np.array([0.3, 0.0, 0.0])
    for _ in range(4):
        rs, rm = single.find_best_action(pose, vel), multi.find_best_action(pose, vel)
        assert rs[0] == rm[0] and rs[2] == rm[2] == BRANCH_GRID and np.array_equal(rs[1], rm[1])
        assert np.array_equal(single.last_costs(), multi.last_costs())
        cmd = rs[1]
        pose = pose + np.array([cmd[0] * math.cos(pose[2]) * 0.3, cmd[0] * math.sin(pose[2]) * 0.3, cmd[2] * 0.3])
        vel = np.array([cmd[0], 0.0, cmd[2]])
    assert multi.ranks() == len(devices)
    a1, n1 = single.all_trajectories(len(lin) * len(ang), 40)
    a2, n2 = multi.all_trajectories(len(lin) * len(ang), 40)
    assert np.array_equal(n1, n2) and np.array_equal(a1, a2)
    idx = int(np.flatnonzero(multi.last_costs() >= 0)[-1])
    assert np.array_equal(single.trajectory_points(idx), multi.trajectory_points(idx))


def test_cpp_multi_device_example():
    """examples/multi_device.cpp: a C++ caller of sfw_multi_* (R = 1 over RCCL, R = 2 host reduce) against
    sfw_score_grid on the same inputs."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "social_force_window_planner_amd", "csrc"), "multidemo"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([os.path.join(ROOT, "build", "multi_device")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl R=1: identical" in r.stdout and "host-reduce R=2: identical" in r.stdout


def test_the_rccl_that_was_resolved_is_reported():
    """which librccl the one-process path binds is stated, not guessed.  Three fresh processes:
    (a) plain: the loader's search, the path ncclAllReduce came from is reported;
    (b) a process that has torch (and with it torch's bundled RCCL) mapped reuses THAT copy — no second RCCL beside it;
    (c) SFW_RCCL_LIB names a file: honoured first; a file that cannot be loaded is an error, not a silent fall-through."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, os, sys
sys.path.insert(0, os.getcwd())
if os.environ.get("WITH_TORCH") == "1":
    import torch
    torch.cuda.init()
from social_force_window_planner_amd import planner
from social_force_window_planner_amd._abi import SFW_MULTI_RCCL, default_params
try:
    m = planner.MultiScorer(default_params(), devices=(0,), exchange=SFW_MULTI_RCCL)
    d = m.describe()
    m.close()
    maps = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
    print(json.dumps({"path": d["rccl_path"], "found": d["rccl_found"], "version": d["rccl_version"], "mapped": sorted(set(maps))}))
except Exception as e:
    print(json.dumps({"error": str(e)}))
'''

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("SFW_RCCL_LIB", "WITH_TORCH")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=e, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert lines, r.stderr[-2000:]
        return json.loads(lines[-1])

    a = run()
    assert "error" not in a and os.path.isfile(a["path"]) and "librccl" in a["path"] and a["version"] > 0, a
    assert a["found"] in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"), a
    b = run(WITH_TORCH="1")
    assert "error" not in b and b["found"] == "already mapped", b
    assert "torch" in b["path"] and len(b["mapped"]) == 1 and os.path.samefile(b["mapped"][0], b["path"]), b
    c = run(SFW_RCCL_LIB=a["path"])
    assert c["found"] == "SFW_RCCL_LIB" and os.path.samefile(c["path"], a["path"]), c
    d = run(SFW_RCCL_LIB="/nonexistent/librccl.so")
    assert "error" in d and "SFW_RCCL_LIB" in d["error"], d
