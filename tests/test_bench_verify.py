"""bench.py's verify leg without a GPU: the launch's cost vector is stood in for by the oracle's own costs of a small grid
(whole-grid branch), and a budget too small for the grid forces the sub-grid + lowest-cost-samples branch that every entry of
`extra` uses for the grids the oracle cannot finish.  Checked: a faithful vector passes both branches, a vector with another
winner or another sentinel set fails them."""
import dataclasses
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle.sfw_oracle import OracleScorer  # noqa: E402
from social_force_window_planner_amd import synthetic as syn  # noqa: E402
from social_force_window_planner_amd._abi import default_params  # noqa: E402


def _job(nv=14, nw=15):
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=nv, nw=nw, sim_time=0.25)
    scene = syn.make_scene(w)
    kw = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
    o = OracleScorer(default_params(**kw))
    o.load_scene(scene)
    costs, best = o.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args, n_threads=8)
    job = types.SimpleNamespace(scene=scene, params_kw=kw, lin=scene.linvels, ang=scene.angvels, lin_all=scene.linvels, index_base=0)
    return job, costs, best


def test_whole_grid_branch():
    job, costs, best = _job()
    v = bench.verify_against_oracle(job, costs, best, 600.0)
    assert v["coverage"].startswith("the whole grid") and v["samples"] == len(costs)
    assert v["cmd_vel_match"] is True and v["invalid_set_equal"] is True and v["max_rel_err"] == 0.0 and v["within_tolerance"] is True
    bad = costs.copy()
    loser = int(np.flatnonzero(bad >= 0)[-1])
    assert loser != best["index"]
    bad[loser] = best["cost"] * 0.5
    v = bench.verify_against_oracle(job, bad, dict(best, index=loser), 600.0)
    assert v["cmd_vel_match"] is False and v["within_tolerance"] is False


def test_sub_grid_branch_with_the_lowest_cost_samples():
    job, costs, best = _job()
    v = bench.verify_against_oracle(job, costs, best, 1e-9, topk=8)  # no budget at all: a 2 x 2 sub-grid + 8 samples
    assert "sub-grid" in v["coverage"] and v["samples"] < len(costs)
    assert v["cmd_vel_match"] is True and "lowest" in v["cmd_vel_scope"] and v["within_tolerance"] is True
    assert v["oracle_cmd_vel"]["index"] == best["index"]
    # a launch that prefers another sample: the oracle's order over the scored candidates names the true best
    bad = costs.copy()
    loser = int(np.flatnonzero(bad >= 0)[-1])
    bad[loser] = best["cost"] * 0.5
    v = bench.verify_against_oracle(job, bad, dict(best, index=loser), 1e-9, topk=8)
    assert v["cmd_vel_match"] is False and v["within_tolerance"] is False
    # a launch that calls its own winner valid where the oracle rejects it
    rej = np.flatnonzero(costs == -1.0)
    if len(rej):
        bad = costs.copy()
        bad[rej[0]] = best["cost"] * 0.5
        v = bench.verify_against_oracle(job, bad, dict(best, index=int(rej[0])), 1e-9, topk=8)
        assert v["cmd_vel_match"] is False and v["invalid_set_equal"] is False and v["within_tolerance"] is False


def test_nothing_selectable():
    job, costs, best = _job(6, 6)
    none = np.full_like(costs, -1.0)
    v = bench.verify_against_oracle(job, none, dict(best, index=-1, n_valid=0), 1e-9, topk=8)
    assert v["invalid_set_equal"] is False  # the oracle has valid samples here; the selection check has no candidate to score
