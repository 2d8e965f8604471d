#!/usr/bin/env python3
"""K2 time of the cfg2 / target grids with and without pedestrian groups (group forces: gaze, coherence, repulsion)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer

for name in ("cfg2", "target"):
    for grouped in (0, 6, 100):
        w = syn.WORKLOADS[name]
        scene = syn.make_scene(w)
        n = w.n_people
        members = min(grouped, n)
        for i in range(1, members + 1):  # groups of three neighbours in index order
            scene.agents[i].group_id = (i - 1) // 3
        g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
        g.set_timing(True)
        g.load_scene(scene)
        g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        ms = []
        for it in range(6):
            g.launch()
            g.fetch(want_costs=False)
            if it >= 2:
                ms.append((g.last_launch_ms(2), g.last_launch_ms(0)))
        k2, tot = np.mean(ms, axis=0)
        print(f"{name}: {members:3d} of {n} pedestrians in groups of 3: K2 {k2:.3f} ms, launch {tot:.3f} ms, "
              f"{w.nv * w.nw / tot / 1e3:.3g}e6 trajectories/s")
