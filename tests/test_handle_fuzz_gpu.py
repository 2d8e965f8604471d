"""One handle driven through a random sequence of world changes and calls (agent counts 0..70, map sizes,
step counts, grids from 2x3 to 72x64 — shared-prefix rollout and, under a 1 MiB table budget, chunked
launches included —, sfw_score_one and Trajectory point queries in between), every result checked
against a fresh oracle.  Guards the handle's buffer management: arenas that grow and shrink, the
class tables, the selection record behind the cost vector."""
import dataclasses
import os

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params

pytestmark = pytest.mark.gpu


def _check(oc, ob, gc, gb):
    assert np.array_equal(oc < 0, gc < 0) and np.array_equal(oc[oc < 0], gc[gc < 0])
    v = oc >= 0
    if v.any():
        assert np.max(np.abs(gc[v] - oc[v]) / np.maximum(np.abs(oc[v]), 1e-300)) <= 1e-9
    assert gb["index"] == ob["index"] and gb["n_valid"] == ob["n_valid"]


@pytest.mark.parametrize("seed,budget", [(1, None), (2, 1), (3, None)])
def test_random_call_sequence(oracle_mod, hip_mod, seed, budget):
    rng = np.random.default_rng(9000 + seed)
    old = os.environ.get("SFW_TABLE_BUDGET_MB")
    try:
        if budget is not None:
            os.environ["SFW_TABLE_BUDGET_MB"] = str(budget)
        g = hip_mod.HipScorer(default_params())
    finally:
        if old is None:
            os.environ.pop("SFW_TABLE_BUDGET_MB", None)
        else:
            os.environ["SFW_TABLE_BUDGET_MB"] = old
    for it in range(10):
        big = it % 4 == 1
        n = int(rng.choice([0, 1, 4, 9, 20, 33, 70]))
        steps = int(rng.choice([1, 6, 20, 40]))
        w = dataclasses.replace(
            syn.WORKLOADS["cfg2"], nv=72 if big else int(rng.integers(2, 9)), nw=64 if big else int(rng.integers(3, 10)),
            n_people=min(n, 12) if big else n, map_size=int(rng.choice([100, 200, 320])), sim_time=steps * 0.025,
            seed=int(rng.integers(0, 10**6)), n_obstacles=int(rng.choice([0, 0, 7])), n_discs=int(rng.integers(0, 15)))
        scene = syn.make_scene(w)
        p = default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity,
                           social_weight=float(rng.uniform(0.5, 2.0)))
        g.set_params(p)
        g.load_scene(scene)
        o = oracle_mod.OracleScorer(p)
        o.load_scene(scene)
        gc, gb = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        if big:
            assert g.plan_info()["samples"] == w.nv * w.nw  # (whether the plan shares a prefix is the heuristic's call)
            rows = rng.choice(w.nv, size=3, replace=False)
            for r in rows:  # oracle on three rows of the large grid
                oc, _ = o.score_grid(scene.robot_state, scene.linvels[r:r + 1], scene.angvels, scene.goal_args, n_threads=16)
                row = gc[r * w.nw:(r + 1) * w.nw]
                v = oc >= 0
                assert np.array_equal(oc < 0, row < 0)
                if v.any():
                    assert np.max(np.abs(row[v] - oc[v]) / np.abs(oc[v])) <= 1e-9
        else:
            oc, ob = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, n_threads=16)
            _check(oc, ob, gc, gb)
        if it % 3 == 0:  # the scalar entry point in between, then the grid again (it clobbers the staged grid)
            vx, vth = float(scene.linvels[-1]), float(scene.angvels[1 % len(scene.angvels)])
            c1, pts1 = g.score_one(scene.robot_state, vx, 0.0, vth, scene.goal_args)
            c0, pts0 = o.score_one(scene.robot_state, vx, 0.0, vth, scene.goal_args)
            assert (c0 < 0) == (c1 < 0) and (c0 < 0 or abs(c1 - c0) <= 1e-9 * abs(c0))
            assert len(pts0) == len(pts1)
            gc2, gb2 = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
            assert np.array_equal(gc, gc2) and gb == gb2
        if it % 3 == 2 and not big:
            _, npts = g.grid_points_batch(0, len(gc), w.n_steps)
            assert np.all(npts[gc >= 0] == w.n_steps)
