"""Round 6's host-side plumbing around the kernels, through the C ABI: the cost vector's pinned mirror (sfw_grid_costs_view), the
polling wait, the world state that is snapshotted by sfw_set_* and sent by the NEXT stage (a small costmap inside the stage's one
arena copy, a large one by its own), the prepared blocking call bench.py times.  Results must not depend on any of it."""
import dataclasses

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64), np.asarray(b, dtype=np.float64).view(np.uint64))


def _params(w):
    return default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity)


@pytest.mark.parametrize("name,nv,nw", [("ref5x9", 5, 9), ("cfg2", 40, 36), ("cfg2", 96, 96)])
def test_the_view_is_the_vector(hip_mod, monkeypatch, name, nv, nw):
    """One-launch control cycle, three-kernel small grid, GPU-filling grid with the shared prefix: the pinned vector the launch
    left equals the copied one; no view before a fetch, after sfw_score_one, or when the mirror is switched off."""
    w = dataclasses.replace(syn.WORKLOADS[name], nv=nv, nw=nw) if name != "ref5x9" else syn.WORKLOADS[name]
    scene = syn.make_scene(w)
    g = hip_mod.HipScorer(_params(w))
    g.load_scene(scene)
    g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    g.launch()
    assert g.costs_view() is None                       # nothing has waited for the launch yet
    costs, best, key = g.fetch()
    view = g.costs_view()
    assert view is not None and not view.flags.writeable and _same(view, costs)
    _, best2, _ = g.fetch(want_costs=False)             # a second fetch of the same launch: same record, same view
    assert best2 == best and _same(g.costs_view(), costs)
    c1, _ = g.score_one(scene.robot_state, 0.2, 0.0, 0.1, scene.goal_args)
    assert g.costs_view() is None                       # score_one clobbers the staged grid
    # the prepared call bench.py times returns that view
    p = g.prepared(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, zero_copy=True)
    for _ in range(3):
        pc, pb, _ = p.step()
        assert _same(pc, costs) and pb == best and not pc.flags.writeable
    # ... and with the mirror off (a device-to-host copy, as rounds 1-5) or the wait blocking at once: the same numbers
    for env in ({"SFW_MIRROR_MAX_MB": "0"}, {"SFW_SPIN_US": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g2 = hip_mod.HipScorer(_params(w))
        g2.load_scene(scene)
        c2, b2 = g2.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        assert _same(c2, costs) and b2 == best
        assert (g2.costs_view() is None) == ("SFW_MIRROR_MAX_MB" in env)
        p2 = g2.prepared(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, zero_copy=True)
        assert _same(p2.step()[0], costs)               # (falls back to the copy when there is no view)
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("map_size", [160, 400])   # 25 KB: rides in the arena copy; 160 KB: a copy of its own
def test_a_costmap_takes_effect_with_the_next_stage(oracle_mod, hip_mod, map_size):
    """sfw_set_costmap snapshots; the NEXT stage sends it.  A launch of a grid staged before the call still scores the old map;
    the stage after it scores the new one — another map of the same size, one of another size, and the first one again."""
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=12, nw=12, n_people=6, map_size=map_size, n_discs=40, seed=77)
    a = syn.make_scene(w)
    def with_patch(scene, value):
        """a block of non-lethal cost ahead of the robot: every forward sample's footprint crosses it"""
        sc = dataclasses.replace(scene, cells=scene.cells.copy())
        ix0, ix1 = (int((x - sc.origin_x) / sc.resolution) for x in (0.2, 1.2))
        iy0, iy1 = (int((y - sc.origin_y) / sc.resolution) for y in (-0.8, 0.8))
        sc.cells[iy0:iy1, ix0:ix1] = value
        return sc

    b = with_patch(a, 120)                                                    # same size, other cells
    c = with_patch(syn.make_scene(dataclasses.replace(w, map_size=map_size + 40, seed=77)), 60)   # another size
    c.agents, c.obstacles = a.agents, a.obstacles

    def oracle(scene):
        o = oracle_mod.OracleScorer(_params(w))
        o.load_scene(scene)
        return o.score_grid(a.robot_state, a.linvels, a.angvels, a.goal_args)[0]

    oa, ob, oc = oracle(a), oracle(b), oracle(c)
    assert not np.array_equal(oa, ob) and not np.array_equal(oa, oc)

    def close(x, y):
        v = y >= 0
        return np.array_equal(x < 0, y < 0) and np.max(np.abs(x[v] - y[v]) / np.abs(y[v])) <= 1e-9

    g = hip_mod.HipScorer(_params(w))
    g.load_scene(a)
    g.stage(a.robot_state, a.linvels, a.angvels, a.goal_args)
    g.set_costmap(b.cells, b.origin_x, b.origin_y, b.resolution)             # after the stage: not this launch's map
    g.launch()
    assert close(g.fetch()[0], oa)
    for scene, want in ((b, ob), (c, oc), (a, oa), (a, oa)):
        g.set_costmap(scene.cells, scene.origin_x, scene.origin_y, scene.resolution)
        got, _ = g.score_grid(a.robot_state, a.linvels, a.angvels, a.goal_args)
        assert close(got, want)
    # an agent set that grows moves the arena: the (unchanged, resident) small map must survive it
    big = syn.make_scene(dataclasses.replace(w, n_people=40, seed=77))
    big.cells, big.origin_x, big.origin_y = a.cells, a.origin_x, a.origin_y
    g.set_agents(big.agents, big.obstacles)
    got, _ = g.score_grid(a.robot_state, a.linvels, a.angvels, a.goal_args)
    o = oracle_mod.OracleScorer(_params(w))
    o.load_scene(big)
    assert close(got, o.score_grid(a.robot_state, a.linvels, a.angvels, a.goal_args)[0])


def test_a_control_cycle_without_any_copy(hip_mod, monkeypatch):
    """A control cycle's grid whose costmap has not changed since the last stage enqueues NO copy: the one-launch kernel's blocks
    fetch the stage's arena (agents, footprint, sample vectors) from pinned memory themselves (sfw_launch.arena_host).  Against
    the same cycles with SFW_ARENA_DIRECT=0 (the arena copied at stage time): a crowd that changes every cycle, a map that changes
    in some cycles and not in others, a second launch of one stage, parameters that change between stage and launch."""
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=7, n_obstacles=20, seed=41)
    scenes = [syn.make_scene(dataclasses.replace(w, seed=41 + q, n_people=7 + (q % 3))) for q in range(6)]
    maps = [scenes[0].cells, scenes[0].cells, scenes[3].cells, scenes[3].cells, scenes[0].cells, scenes[0].cells]

    def run(direct):
        monkeypatch.setenv("SFW_ARENA_DIRECT", "1" if direct else "0")
        g = hip_mod.HipScorer(_params(w))
        g.set_footprint(scenes[0].footprint)
        out = []
        for q, sc in enumerate(scenes):
            g.set_costmap(maps[q], sc.origin_x, sc.origin_y, sc.resolution)
            g.set_agents(sc.agents, sc.obstacles)
            costs, best = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
            assert g.plan_info()["one_launch"] == 1
            out.append((costs.copy(), best))
            if q == 2:       # the same stage launched again: the device copy of the arena is there
                g.launch()
                c2, b2, _ = g.fetch()
                assert _same(c2, costs) and b2 == best
            if q == 4:       # parameters change between stage and launch: the launch re-plans and must not lose the new rest terms
                g.stage(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
                p2 = default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity, social_weight=2.5)
                g.set_params(p2)
                g.launch()
                out.append((g.fetch()[0].copy(), None))
                g.set_params(_params(w))
        return out

    a, b = run(True), run(False)
    assert len(a) == len(b) == 7
    for (ca, ba), (cb, bb) in zip(a, b):
        assert _same(ca, cb) and ba == bb
    assert not _same(a[0][0], a[2][0])


def test_a_points_dump_between_stage_and_launch(hip_mod):
    """sfw_grid_points_batch needs only a staged grid; with the arena still in pinned memory only (a control cycle's stage
    enqueues no copy) the dump must send it first.  Same points as after the launch, and the launch still scores the same."""
    w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=6, seed=52)
    scene = syn.make_scene(w)
    g = hip_mod.HipScorer(_params(w))
    g.load_scene(scene)
    ref_costs, ref_best = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    ref_pts, ref_n = g.grid_points_batch(0, 45, w.n_steps)
    g2 = hip_mod.HipScorer(_params(w))
    g2.load_scene(scene)
    g2.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    pts, n = g2.grid_points_batch(0, 45, w.n_steps)     # before any launch: point counts lack the pedestrian contacts only
    assert _same(pts[:, 0, :], ref_pts[:, 0, :]) and np.all(n >= ref_n)
    g2.launch()
    costs, best, _ = g2.fetch()
    assert _same(costs, ref_costs) and best == ref_best
    pts, n = g2.grid_points_batch(0, 45, w.n_steps)
    assert np.array_equal(n, ref_n) and _same(pts, ref_pts)
