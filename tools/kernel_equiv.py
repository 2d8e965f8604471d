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
for n in (1, 5, 20, 21, 31, 50, 63, 64, 90, 127):
    w = dataclasses.replace(syn.WORKLOADS["cfg2"], nv=5, nw=7, n_people=n, seed=400 + n)
    sc = syn.make_scene(w)
    g = HipScorer(default_params()); g.load_scene(sc)
    c, b = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
    out[str(n)] = c
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
    print(f"N={n:>4}: flat==reg {np.array_equal(f, r)}  default==flat {np.array_equal(a, f)}  max|f-r| {np.abs(f - r).max():.2e}")
