#!/usr/bin/env python3
"""K2 time of the two wave organisations over the crowd size on a GPU-filling grid (128 x 128 samples, 40 steps):
where does the register-resident form (lanes used G*A/64, 90 VALU and 2 atomics per pair, six waves per SIMD) stop
beating the flat one (all lanes, 85 VALU and 4 atomics, five waves)?  Input to plan_for()'s score.
usage: form_crossover.py [agent counts ...]"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import SFW_K2_FLAT, SFW_K2_REGISTER, default_params
from social_force_window_planner_amd.planner import HipScorer

counts = [int(a) for a in sys.argv[1:]] or [12, 16, 21, 22, 25, 27, 30, 32, 33, 36, 40, 43, 46, 49, 51, 55, 58, 61, 64]
for A in counts:
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], n_people=A - 1, seed=300 + A)
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
    G = min(64 // A, 32)
    P = A * (A - 1) // 2
    fill = P / (64.0 * ((P + 63) // 64))
    print(f"A={A:3d}  register (G={G}, lanes {G * A / 64:.2f}) {out[0]:.3f} ms   flat (fill {fill:.2f}) {out[1]:.3f} ms   "
          f"flat/register {out[1] / out[0]:.3f}   lanes/fill {G * A / 64 / fill:.3f}")
