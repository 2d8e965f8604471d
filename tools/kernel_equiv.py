#!/usr/bin/env python3
"""Are the K2 organisations bit-identical?  Scores the same scene with the default plan and
with SFW_FORCE_FLAT=1 / 0 (separate processes: the override is read once per process)."""
import dataclasses, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, dataclasses, numpy as np
sys.path.insert(0, %r)
from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer
out = {}
for n, n_obs, grouped in [(1, 0, 0), (5, 0, 0), (20, 0, 0), (21, 0, 0), (31, 0, 0), (50, 0, 0), (63, 0, 0), (64, 0, 0),
                          (90, 0, 0), (127, 0, 0), (5, 60, 0), (20, 64, 0), (50, 7, 0), (50, 64, 0), (63, 1, 0), (100, 33, 0),
                          (20, 0, 1), (50, 16, 1), (100, 9, 1)]:
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=5, nw=7, n_people=n, seed=400 + n, n_obstacles=n_obs)
    sc = syn.make_scene(w)
    if grouped:
        for i in range(1, min(n, 13)):
            sc.agents[i].group_id = i // 3
    g = HipScorer(default_params()); g.load_scene(sc)
    c, b = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
    out[f"{n}_o{n_obs}_g{grouped}"] = c
np.savez(sys.argv[1], **out)
''' % ROOT
import numpy as np
res = {}
for tag, env in (("default", {}), ("flat", {"SFW_FORCE_FLAT": "1"}), ("reg", {"SFW_FORCE_FLAT": "0"})):
    e = dict(os.environ); e.update(env)
    path = f"/tmp/keq_{tag}.npz"
    subprocess.run([sys.executable, "-c", CHILD, path], check=True, env=e)
    res[tag] = np.load(path)
for n in res["default"].files:
    a, f, r = res["default"][n], res["flat"][n], res["reg"][n]
    print(f"N={n:>12}: flat==reg {np.array_equal(f, r)}  default==flat {np.array_equal(a, f)}  max|f-r| {np.abs(f - r).max():.2e}")
