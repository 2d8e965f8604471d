#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with the CPU oracle.

Each .npz is pure data: the complete inputs of one score_grid call (costmap,
footprint, agents, obstacle points, robot state, sample vectors, parameters)
and the expected outputs (per-sample cost vector, selected command).  There is
no upstream golden data for this path (the reference ships no tests, SURVEY.md
§4), so the expected values are the oracle's; the pieces of the oracle that CAN
be checked against real reference code are checked in tests/test_oracle_kat.py.

Run from the repo root:  python tests/golden/make_golden.py
"""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.sfw_oracle import OracleScorer  # noqa: E402
from social_force_window_planner_amd import synthetic as syn  # noqa: E402
from social_force_window_planner_amd._abi import SfwAgent, default_params  # noqa: E402

AGENT_FIELDS = [n for n, _ in SfwAgent._fields_]
PARAM_FIELDS = ["max_vel_x", "sim_time", "sim_granularity", "robot_radius", "social_weight", "costmap_weight",
                "angle_weight", "distance_weight", "vel_weight"]


def agents_to_array(agents):
    return np.array([[float(getattr(a, f)) for f in AGENT_FIELDS] for a in agents], dtype=np.float64).reshape(
        len(agents), len(AGENT_FIELDS))


def blocked_scene(seed):
    """Robot 0.9 m from a lethal disc and near the unknown border: many rejected
    trajectories (all three rejection causes occur)."""
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=9, nw=11, n_people=6, seed=seed, map_size=120)
    sc = syn.make_scene(w)
    res = sc.resolution
    ys, xs = np.meshgrid(sc.origin_y + (np.arange(120) + 0.5) * res, sc.origin_x + (np.arange(120) + 0.5) * res,
                         indexing="ij")
    d = np.hypot(xs - 0.95, ys - 0.25)
    val = np.where(d <= 0.15, 254.0, np.clip(253.0 * (1 - (d - 0.15) / 0.5), 0, 253))
    sc.cells[:] = np.maximum(sc.cells, val.astype(np.uint8))
    # a person that will be run over by the fast straight samples
    sc.agents[1].x, sc.agents[1].y, sc.agents[1].vx, sc.agents[1].vy = 0.55, -0.45, 0.0, 0.35
    sc.agents[1].goal_x, sc.agents[1].goal_y = 0.55, 0.25
    return sc


def grouped_scene(seed):
    """Two walking groups (3 and 2 members, distinct speeds: exact relative rest is the one input class on
    which the reference's sign(theta) is rounding noise), a singleton group, 6 laser points."""
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=6, nw=7, n_people=11, seed=seed, n_obstacles=6)
    sc = syn.make_scene(w)
    ag = sc.agents
    for gid, members in ((3, (1, 2, 3)), (8, (5, 6))):
        lead = ag[members[0]]
        for k, m in enumerate(members):
            ag[m].group_id = gid
            ag[m].x, ag[m].y = lead.x + 0.45 * k, lead.y + 0.3 * k
            ag[m].vx, ag[m].vy = lead.vx * (1.0 + 0.03 * k) + 0.01 * k, lead.vy * (1.0 - 0.02 * k)
            ag[m].goal_x, ag[m].goal_y = ag[m].x + 2.0 * ag[m].vx, ag[m].y + 2.0 * ag[m].vy
    ag[9].group_id = 40  # a group of one: no group force
    return sc


def standing_scene(seed):
    """Exact relative rest (round 2): three standing people (v = 0,0 — what a tracker reports for static persons), a
    standing pair inside a group, two walkers with bit-identical velocities.  lightsfm's sign(theta) for such pairs is the
    rounding of two atan2; the fixture pins what the oracle (this image's libm) makes of it."""
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=8, nw=9, n_people=12, seed=seed)
    sc = syn.make_scene(w)
    ag = sc.agents
    for i in (1, 2, 3, 4, 5):
        ag[i].vx = ag[i].vy = 0.0
        ag[i].goal_x, ag[i].goal_y = ag[i].x, ag[i].y
    for i in (4, 5):
        ag[i].group_id = 7
    ag[5].x, ag[5].y = ag[4].x + 0.5, ag[4].y + 0.3
    ag[5].goal_x, ag[5].goal_y = ag[5].x, ag[5].y
    ag[7].vx, ag[7].vy = ag[6].vx, ag[6].vy
    ag[7].goal_x, ag[7].goal_y = ag[7].x + 2.0 * ag[7].vx, ag[7].y + 2.0 * ag[7].vy
    return sc


def collinear_scene(seed):
    """Motion exactly along the connecting line (round 3): grid-aligned people walking towards each other (lightsfm's theta
    is rounding noise around 0, like relative rest) and apart faster than 1 / lambda (theta = +-pi), the robot on the line of
    two of them, one pair on a diagonal.  w x diff == 0 for all these pairs; the fixture pins what the oracle (this image's
    libm) makes of them."""
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=8, nw=9, n_people=14, seed=seed)
    sc = syn.make_scene(w)

    def put(i, x, y, vx, vy):
        a = sc.agents[i]
        a.x, a.y, a.vx, a.vy = x, y, vx, vy
        a.goal_x, a.goal_y = x + 2.0 * vx, y + 2.0 * vy

    put(1, 2.0, 1.0, -0.8, 0.0)
    put(2, 4.0, 1.0, 0.8, 0.0)
    put(3, -2.0, -1.5, 0.6, 0.0)
    put(4, -0.5, -1.5, -0.6, 0.0)
    put(5, -3.0, 2.0, 0.0, 0.7)
    put(6, -3.0, 0.5, 0.0, -0.9)
    put(7, 3.0, 0.0, -0.9, 0.0)
    put(8, -2.5, 0.0, -1.0, 0.0)
    put(9, 1.5, 2.5, 0.25, 0.25)
    put(10, 2.5, 3.5, 0.75, 0.75)
    return sc


def cases():
    yield "cfg1", syn.make_scene("cfg1"), {}
    for n in (0, 1, 5):
        w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=n, seed=70 + n)
        yield f"ref5x9_n{n}", syn.make_scene(w), {}
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=12, nw=12, seed=21, n_obstacles=8)
    yield "cfg2_12x12_obs8", syn.make_scene(w), {}
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=6, nw=7, n_people=70, seed=22, footprint="point")
    yield "crowd70_point", syn.make_scene(w), {}
    yield "blocked", blocked_scene(23), {}
    yield "groups_obs", grouped_scene(25), {}
    yield "standing_people", standing_scene(501), {}
    yield "collinear_walkers", collinear_scene(611), {}
    w = dataclasses.replace(syn.WORKLOADS["cfg3"], nv=5, nw=6, map_size=200, seed=24)
    yield "cfg3_5x6_yamlweights", syn.make_scene(w), dict(social_weight=2.0, vel_weight=0.8, angle_weight=0.6,
                                                          max_vel_x=0.8, robot_radius=0.4)


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])  # optional: regenerate just the named fixtures
    for name, sc, pkw in cases():
        if only and name not in only:
            continue
        w = sc.workload
        p = default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity, **pkw)
        o = OracleScorer(p)
        o.load_scene(sc)
        costs, best = o.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            cells=sc.cells, origin=np.array([sc.origin_x, sc.origin_y, sc.resolution]), footprint=sc.footprint,
            agents=agents_to_array(sc.agents), agent_fields=np.array(AGENT_FIELDS), obstacles=sc.obstacles,
            robot_state=np.array(sc.robot_state), goal_args=np.array(sc.goal_args), linvels=sc.linvels,
            angvels=sc.angvels, params=np.array([float(getattr(p, f)) for f in PARAM_FIELDS]),
            param_fields=np.array(PARAM_FIELDS), costs=costs,
            best=np.array([best["index"], best["cost"], best["vx"], best["vtheta"], best["n_valid"]]))
        print(f"{name:24s} T={len(costs):4d} invalid={(costs == -1).sum():3d} best={best['index']} "
              f"cost={best['cost']:.6f}")


if __name__ == "__main__":
    main()
