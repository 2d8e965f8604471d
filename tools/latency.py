#!/usr/bin/env python3
"""Control-cycle latency of the reference's own configuration: 5x9 samples,
N pedestrians, full blocking call incl. world upload (what findBestAction pays)."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer
from oracle.sfw_oracle import OracleScorer
for n in (0, 5, 20, 50):
    for st, gr in ((1.0, 0.025), (1.5, 0.25)):
        w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=n, sim_time=st, sim_granularity=gr)
        sc = syn.make_scene(w)
        p = default_params(sim_time=st, sim_granularity=gr)
        g = HipScorer(p); g.load_scene(sc)
        o = OracleScorer(default_params(sim_time=st, sim_granularity=gr)); o.load_scene(sc)
        for _ in range(5): g.load_scene(sc); g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter(); g.load_scene(sc); g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(5): o.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
        tc = (time.perf_counter() - t0) / 5
        print(f"N={n:3d} S={w.n_steps:3d}: GPU cycle median {np.median(ts)*1e3:7.3f} ms (min {min(ts)*1e3:.3f}), CPU oracle 1 thread {tc*1e3:8.3f} ms")
