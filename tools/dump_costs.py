"""Dump the per-sample costs of a few workloads to gpurun_out/costs_<tag>.npz, or compare the current
build against such a dump (bitwise).  usage: dump_costs.py dump|compare <tag>"""
import dataclasses
import sys

import numpy as np

from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import default_params
from social_force_window_planner_amd.planner import HipScorer

CASES = {
    "cfg2": syn.WORKLOADS["cfg2"],
    "target64": dataclasses.replace(syn.WORKLOADS["target"], nv=64, nw=64),
    "target_small": dataclasses.replace(syn.WORKLOADS["target"], nv=16, nw=16),  # T <= 4096: flat kernel
    "cfg2_small": dataclasses.replace(syn.WORKLOADS["cfg2"], nv=32, nw=32),
    "cfg5_32": dataclasses.replace(syn.WORKLOADS["cfg5"], nv=96, nw=96),  # A = 101: NS = 2 register kernel
    "cfg4_16": dataclasses.replace(syn.WORKLOADS["cfg4"], nv=16, nw=16),
    "obst": dataclasses.replace(syn.WORKLOADS["cfg2"], nv=24, nw=24, n_obstacles=12),
    "contacts": dataclasses.replace(syn.WORKLOADS["cfg4"], nv=8, nw=8, people_r_in=0.8),
}


def run(w):
    scene = syn.make_scene(w)
    p = default_params()
    p.sim_time = w.sim_time
    p.sim_granularity = w.sim_granularity
    g = HipScorer(p)
    g.load_scene(scene)
    costs, _ = g.score_grid(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    return np.asarray(costs)


def main():
    mode, tag = sys.argv[1], sys.argv[2]
    path = f"gpurun_out/costs_{tag}.npz"
    out = {k: run(w) for k, w in CASES.items()}
    if mode == "dump":
        np.savez_compressed(path, **out)
        print("dumped", path)
        return
    ref = np.load(f"build/ref/costs_{tag}.npz")
    for k, c in out.items():
        r = ref[k]
        same = np.array_equal(c.view(np.uint64), r.view(np.uint64))
        valid = (r >= 0) & (c >= 0)
        rel = np.max(np.abs(c[valid] - r[valid]) / np.maximum(np.abs(r[valid]), 1e-300)) if valid.any() else 0.0
        print(f"{k:14s} bitwise {'SAME' if same else 'DIFF'}  status-match {np.array_equal(r < 0, c < 0)}  max rel {rel:.3e}")


main()
