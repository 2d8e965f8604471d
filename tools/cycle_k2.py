#!/usr/bin/env python3
"""K2 (and whole-launch) time of the reference's own 5 x 9 control-cycle grid for N people and O laser points: the latency-bound
end of the path (a lone wave per SIMD).  Honours SFW_HIP_LIB, so builds can be compared inside one GPU session.
usage: cycle_k2.py [N,N,...] [O,O,...]"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer

Ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,5,20,50").split(",")]
Os = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,60,240,720").split(",")]
for n in Ns:
    row = []
    for o in Os:
        w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=n, n_obstacles=o, seed=900 + n)
        scene = syn.make_scene(w)
        g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
        g.set_timing(True)
        g.load_scene(scene)
        g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
        ms = []
        for it in range(40):
            g.launch()
            g.fetch(want_costs=False)
            if it >= 10:
                ms.append((g.last_launch_ms(2), g.last_launch_ms(0)))
        k2, tot = np.median(ms, axis=0)
        row.append(f"O={o:3d}: K2 {k2 * 1e3:6.1f} us launch {tot * 1e3:6.1f} us")
    print(f"N={n:2d}  " + " | ".join(row))
