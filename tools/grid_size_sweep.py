#!/usr/bin/env python3
"""Throughput of the blocking scoring call over the grid size (nv x nv samples) for the cfg2 and target crowds: where
does the GPU fill, and does any size fall off the curve?  usage: grid_size_sweep.py [workload ...]"""
import dataclasses
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer

for name in sys.argv[1:] or ["cfg2", "target"]:
    for nv in (16, 24, 32, 48, 64, 72, 80, 96, 112, 128, 160, 192, 256):
        w = dataclasses.replace(syn.WORKLOADS[name], nv=nv, nw=nv)
        scene = syn.make_scene(w)
        g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
        g.set_timing(True)
        g.load_scene(scene)
        ts, k2 = [], []
        for it in range(8):
            t0 = time.perf_counter()
            g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
            ts.append(time.perf_counter() - t0)
            k2.append(g.last_launch_ms(2))
        info = g.plan_info()
        t = float(np.median(ts[3:]))
        print(f"{name}: {nv:3d} x {nv:3d} = {nv * nv:6d} samples  call {t * 1e3:7.3f} ms  K2 {np.median(k2[3:]):7.3f} ms  "
              f"{nv * nv / t / 1e6:6.2f}e6 traj/s  levels {info['levels']} organisation {info['organisation']}")
        g.close()
