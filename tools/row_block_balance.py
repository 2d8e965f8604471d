#!/usr/bin/env python3
"""K2 time of every rank's block of a grid cut R ways, equal row counts against sfw_plan_row_blocks — measured one block
after the other on ONE GPU (what each of R GPUs would be busy for).  usage: row_block_balance.py [workload] [R]"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer, plan_row_blocks

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = syn.WORKLOADS[name]
scene = syn.make_scene(dataclasses.replace(w, nv=2, nw=2))
lin, ang = syn.generalised_sampler(w.nv, w.nw)
g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
g.set_timing(True)
g.load_scene(scene)
cuts = {"equal": [r * w.nv // R for r in range(R + 1)],
        "planned": list(plan_row_blocks(lin, ang, scene.robot_state, scene.goal_args, w.sim_time, w.n_steps, w.n_people + 1, R))}
for kind, c in cuts.items():
    ms = []
    for r in range(R):
        g.stage(scene.robot_state, lin[c[r]:c[r + 1]], ang, scene.goal_args, c[r] * w.nw)
        g.launch(); g.fetch(want_costs=False)
        g.launch(); g.fetch(want_costs=False)
        ms.append(g.last_launch_ms(2))
    ms = np.array(ms)
    print(f"{name} R={R} {kind:8s} rows {np.diff(c).tolist()}  K2 ms {np.round(ms, 1).tolist()}  max/mean {ms.max() / ms.mean():.4f}")
