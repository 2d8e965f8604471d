"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces them (guards the checker against drift).
GPU: the HIP path, through the C ABI, reproduces them.
Tolerances: RTOL_F64 = 1e-9 on valid costs (libm last-ulp + summation order);
sentinel sets, the selected index and cmd_vel are exact."""
import numpy as np
import pytest

import golden_util as gu

RTOL_F64 = 1e-9


def _check(fx, costs, best, rtol):
    assert costs.shape == fx.costs.shape
    assert np.array_equal(costs < 0, fx.costs < 0)
    assert np.array_equal(costs[costs < 0], fx.costs[fx.costs < 0])
    v = fx.costs >= 0
    if v.any():
        rel = np.abs(costs[v] - fx.costs[v]) / np.abs(fx.costs[v])
        assert rel.max() <= rtol, f"{fx.name}: max rel err {rel.max():.3e}"
    assert best["index"] == fx.best["index"]
    assert best["vx"] == fx.best["vx"] and best["vtheta"] == fx.best["vtheta"]
    assert best["n_valid"] == fx.best["n_valid"]


def test_fixture_inventory():
    assert {"cfg1", "ref5x9_n0", "ref5x9_n1", "ref5x9_n5", "cfg2_12x12_obs8", "crowd70_point", "blocked",
            "cfg3_5x6_yamlweights", "groups_obs", "standing_people", "collinear_walkers"} <= set(gu.names())


@pytest.mark.parametrize("name", gu.names())
def test_oracle_reproduces_golden(oracle_mod, name):
    fx = gu.Fixture(name)
    o = oracle_mod.OracleScorer(fx.params())
    fx.load_into(o)
    costs, best = o.score_grid(fx.robot_state, fx.linvels, fx.angvels, fx.goal_args)
    _check(fx, costs, best, 1e-12)
    # the OpenMP variant used as the all-cores CPU baseline is bit-identical
    costs_mt, best_mt = o.score_grid(fx.robot_state, fx.linvels, fx.angvels, fx.goal_args, n_threads=4)
    assert np.array_equal(costs, costs_mt) and best == best_mt


@pytest.mark.gpu
@pytest.mark.parametrize("name", gu.names())
def test_hip_reproduces_golden(hip_mod, name):
    fx = gu.Fixture(name)
    g = hip_mod.HipScorer(fx.params())
    fx.load_into(g)
    costs, best = g.score_grid(fx.robot_state, fx.linvels, fx.angvels, fx.goal_args)
    _check(fx, costs, best, RTOL_F64)
    # score_one == the grid entry, for a valid and (if any) an invalid sample
    nw = len(fx.angvels)
    picks = [int(np.flatnonzero(fx.costs >= 0)[0])]
    if (fx.costs == -1).any():
        picks.append(int(np.flatnonzero(fx.costs == -1)[0]))
    for i in picks:
        c, pts = g.score_one(fx.robot_state, fx.linvels[i // nw], 0.0, fx.angvels[i % nw], fx.goal_args)
        if fx.costs[i] < 0:
            assert c == -1.0
        else:
            assert c == pytest.approx(fx.costs[i], rel=RTOL_F64)
