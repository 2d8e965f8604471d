"""A controller runs for hours at 10-20 Hz (nav2 controller_server calling computeVelocityCommands,
/root/reference/src/sfw_planner_node.cpp:220): the handle must neither leak device memory per cycle nor per
create/destroy, whatever sequence of crowd sizes, laser-point counts, grid sizes and entry points it sees.

Device memory is read through hipMemGetInfo (torch.cuda.mem_get_info: plumbing only); the library's buffers only ever
grow to the largest request, so after a warm-up lap over every shape the free figure must not move any more.
"""
import dataclasses
import os

import numpy as np
import pytest

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import SFW_MULTI_HOST_REDUCE, default_params

pytestmark = pytest.mark.gpu

SLACK = 4 << 20  # the runtime's own pools move by whole 2 MiB pages


def _free_bytes():
    import torch

    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def _rss_bytes():
    with open(f"/proc/{os.getpid()}/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def _scenes():
    base = syn.WORKLOADS["cfg2"]
    shapes = [(5, 9, 5, 0), (24, 20, 20, 16), (96, 96, 12, 0), (12, 12, 50, 64), (5, 9, 0, 0), (40, 33, 70, 7)]
    out = []
    for q, (nv, nw, n_people, n_obs) in enumerate(shapes):
        w = dataclasses.replace(base, name=f"life{q}", nv=nv, nw=nw, n_people=n_people, n_obstacles=n_obs, seed=300 + q)
        out.append(syn.make_scene(w))
    return out


def _cycle(g, sc, with_points):
    g.load_scene(sc)
    costs, best = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
    if with_points:
        g.grid_points_batch(0, min(costs.size, 64), sc.workload.n_steps)
    g.score_one(sc.robot_state, 0.2, 0.0, 0.1, sc.goal_args)
    return costs, best


def test_cycles_do_not_leak(hip_mod):
    scenes = _scenes()
    g = hip_mod.HipScorer(default_params(sim_time=1.0, sim_granularity=0.025))
    g.set_timing(True)
    first = []
    for lap in range(2):  # warm-up: every buffer reaches its largest size, the runtime's pools settle
        for q, sc in enumerate(scenes):
            g.set_points_capture(q % 2 == 0)
            c, b = _cycle(g, sc, True)
            if lap == 0:
                first.append((c.copy(), b["index"]))
    free0, rss0 = _free_bytes(), _rss_bytes()
    for lap in range(40):
        for q, sc in enumerate(scenes):
            g.set_points_capture((q + lap) % 2 == 0)
            c, b = _cycle(g, sc, lap % 4 == 0)
            if lap == 39:  # and the 250th cycle still returns what the first did
                assert np.array_equal(c, first[q][0]) and b["index"] == first[q][1]
    free1, rss1 = _free_bytes(), _rss_bytes()
    assert free0 - free1 <= SLACK, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 240 cycles"
    assert rss1 - rss0 <= 16 << 20, f"host RSS grew by {(rss1 - rss0) / 2**20:.1f} MiB over 240 cycles"
    g.close()


def test_create_destroy_does_not_leak(hip_mod):
    scenes = _scenes()

    def one(multi):
        sc = scenes[2] if multi else scenes[1]
        if multi:
            m = hip_mod.MultiScorer(default_params(sim_time=1.0, sim_granularity=0.025), devices=(0, 0, 0),
                                    exchange=SFW_MULTI_HOST_REDUCE)
            m.load_scene(sc)
            m.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
            m.close()
        else:
            g = hip_mod.HipScorer(default_params(sim_time=1.0, sim_granularity=0.025))
            _cycle(g, sc, True)
            g.close()

    for multi in (False, True, False, True):  # warm-up
        one(multi)
    free0, rss0 = _free_bytes(), _rss_bytes()
    rss = [rss0]
    for q in range(40):
        one(q % 2 == 1)
        rss.append(_rss_bytes())
    free1, rss1 = _free_bytes(), rss[-1]
    assert free0 - free1 <= SLACK, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 40 handles"
    # A leak per handle shows as growth at (nearly) every handle.  One step — the HIP runtime growing a pool of its own once,
    # the kernel collapsing the process's pages into huge pages: seen as +189.4 MiB at ONE handle on some boxes, 0 on others
    # with the same library (round 6) — is not a leak of this library's: at most two handles may grow the process at all.
    steps = [b - a for a, b in zip(rss, rss[1:])]
    growing = [i for i, d in enumerate(steps) if d > (256 << 10)]
    assert rss1 - rss0 <= 16 << 20 or len(growing) <= 2, \
        f"host RSS grew by {(rss1 - rss0) / 2**20:.1f} MiB over 40 handles, at handles {growing}"
