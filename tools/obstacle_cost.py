#!/usr/bin/env python3
"""K2 time of the cfg2 / target grids with O laser points (the obstacle term is A x O per step, the pair term A^2/2):
is the cost per point constant, or does the points' LDS footprint (16 B each, per wave) cost occupancy?
usage: obstacle_cost.py [workload ...]"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer

prev = None
for name, n_obs in [(n, o) for n in (sys.argv[1:] or ["cfg2", "target"]) for o in (0, 16, 64, 128, 256, 512, 720)]:
    w = dataclasses.replace(syn.WORKLOADS[name], n_obstacles=n_obs)
    scene = syn.make_scene(w)
    g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
    g.set_timing(True)
    g.load_scene(scene)
    g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    ms = []
    for it in range(8):
        g.launch()
        g.fetch(want_costs=False)
        if it >= 3:
            ms.append((g.last_launch_ms(2), g.last_launch_ms(0)))
    k2, tot = np.mean(ms, axis=0)
    if n_obs == 0:
        base = k2
    per = (k2 - base) / n_obs * 1e3 if n_obs else 0.0
    print(f"{name}: O = {n_obs:4d} laser points: K2 {k2:.3f} ms ({per:.2f} us per point), launch {tot:.3f} ms, {w.nv * w.nw / tot / 1e3:.3g}e6 trajectories/s")
