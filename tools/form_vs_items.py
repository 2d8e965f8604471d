#!/usr/bin/env python3
"""K2 time of the two wave organisations over the ITEM count of a launch (A agents, nv x 128 samples, 40 steps, shared
prefix off so that one launch carries all items): where should plan_for() stop preferring the flat form's shorter critical
path?  usage: SFW_PREFIX=0 form_vs_items.py [A ...]"""
import dataclasses
import os
import sys

if "--prefix" not in sys.argv:
    os.environ.setdefault("SFW_PREFIX", "0")
else:
    sys.argv.remove("--prefix")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import SFW_K2_FLAT, SFW_K2_REGISTER, default_params
from social_force_window_planner_amd.planner import HipScorer

for A in [int(a) for a in sys.argv[1:]] or [6, 11, 21, 31]:
    for nv in ((16, 24, 32, 40, 48) if "SFW_PREFIX" not in os.environ else (4, 8, 16, 24, 32, 40, 48, 64, 96, 128)):
        w = dataclasses.replace(syn.WORKLOADS["cfg2"], n_people=A - 1, seed=300 + A, nv=nv, nw=128)
        scene = syn.make_scene(w)
        out = []
        for form in (SFW_K2_REGISTER, SFW_K2_FLAT):
            g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
            g.set_timing(True)
            g.set_k2_form(form)
            g.load_scene(scene)
            g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
            ms = []
            for it in range(7):
                g.launch()
                g.fetch(want_costs=False)
                if it >= 2:
                    ms.append(g.last_launch_ms(2))
            out.append(float(np.median(ms)))
            g.close()
        print(f"A={A:3d} items={nv * 128:6d}  register {out[0]:.4f} ms   flat {out[1]:.4f} ms   flat/register {out[1] / out[0]:.3f}")
