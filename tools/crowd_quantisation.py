#!/usr/bin/env python3
"""VERDICT r4 item 6, priced before building: would taking the ROBOT's laser-point evaluation out of K2 (into K1: it depends
on the robot's pre-step pose only) shorten the laser-point pass?  The flat form's task loop walks 16 agents x 16 segments per
round, four agents per lane; the target crowd's 51 agents are 3 full rounds + a rest round with one agent per lane, and so
are 50.  This times K2 of the target grid + 720 points for crowds around that size: if A = 50 costs what A = 51 costs, the
robot's evaluation rides in lanes that would idle otherwise and moving it buys nothing.
usage: crowd_quantisation.py [n_people ...]"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer

for rep in range(2):
    for n in [int(v) for v in sys.argv[1:]] or [46, 47, 48, 49, 50, 51]:
        w = dataclasses.replace(syn.WORKLOADS["target_o720"], n_people=n)
        scene = syn.make_scene(w)
        g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
        g.set_timing(True)
        g.load_scene(scene)
        g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        ms = []
        for it in range(6):
            g.launch()
            g.fetch(want_costs=False)
            if it >= 2:
                ms.append(g.last_launch_ms(2))
        A = n + 1
        rounds, rest = divmod(A, 16)
        print(f"A = {A:2d} agents ({rounds} full rounds + {(rest + 3) // 4} agent(s) per lane in the rest round): K2 {np.mean(ms):7.3f} ms, "
              f"{np.mean(ms) / (A * 720) * 1e3:.4f} us per (agent, point) of a step's 40", flush=True)
