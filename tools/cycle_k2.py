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


def launch_us(scene, w, fused):
    """Median device time of the whole scoring launch (HIP events, first kernel's start to the selection's end)."""
    os.environ["SFW_CYCLE_FUSED"] = "1" if fused else "0"   # (read by the library at every launch)
    g = HipScorer(default_params(sim_time=w.sim_time, sim_granularity=w.sim_granularity))
    g.set_timing(True)
    g.load_scene(scene)
    g.stage(scene.robot_state, scene.linvels, scene.angvels, scene.goal_args)
    one = g.plan_info().get("one_launch", 0)
    ms = []
    for it in range(40):
        g.launch()
        g.fetch(want_costs=False)
        if it >= 10:
            ms.append((g.last_launch_ms(2), g.last_launch_ms(0)))
    k2, tot = np.median(ms, axis=0)
    g.close()
    return k2 * 1e3, tot * 1e3, one


print("# device time of a control cycle's scoring launch in us, three kernels (K1 small + K2 flat + K3: rounds 1-5) -> one kernel")
print("# (sfw_cycle_kernel, round 6); the three-kernel cell also shows its K2 alone")
for n in Ns:
    row = []
    for o in Os:
        w = dataclasses.replace(syn.WORKLOADS["ref5x9"], n_people=n, n_obstacles=o, seed=900 + n)
        scene = syn.make_scene(w)
        k2_3, tot_3, _ = launch_us(scene, w, False)
        _, tot_1, one = launch_us(scene, w, True)
        row.append(f"O={o:3d}: {tot_3:6.1f} (K2 {k2_3:6.1f}) -> {tot_1:6.1f}{'' if one else ' (three kernels)'}")
    print(f"N={n:2d}  " + " | ".join(row))
