#!/usr/bin/env python3
"""Measures the f32-forces mode against the CPU oracle: max/median relative cost
error, invalid-set equality, selected-command equality (run on the GPU box)."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.sfw_oracle import OracleScorer
from social_force_window_planner_amd import synthetic as syn
from social_force_window_planner_amd._abi import SFW_PRECISION_F32, default_params
from social_force_window_planner_amd.planner import HipScorer

cases = [("cfg2", dict()), ("target", dict(nv=64, nw=64)), ("cfg3", dict(nv=32, nw=32)), ("cfg4", dict(nv=6, nw=6)),
         ("cfg2", dict(seed=11, n_obstacles=32)), ("cfg2", dict(seed=12, n_people=63)), ("ref5x9", dict())]
for name, kw in cases:
    w = dataclasses.replace(syn.WORKLOADS[name], **kw)
    sc = syn.make_scene(w)
    pk = dict(sim_time=w.sim_time, sim_granularity=w.sim_granularity)
    o = OracleScorer(default_params(**pk)); o.load_scene(sc)
    oc, ob = o.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args, n_threads=os.cpu_count())
    for prec in (0, 1):
        g = HipScorer(default_params(precision=prec, **pk)); g.set_timing(True); g.load_scene(sc)
        gc, gb = g.score_grid(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args)
        t0 = time.perf_counter(); g.stage(sc.robot_state, sc.linvels, sc.angvels, sc.goal_args); g.launch(); g.sync(); 
        ms = g.last_launch_ms(2)
        v = (oc >= 0) & (gc >= 0)
        rel = np.abs(gc[v] - oc[v]) / np.abs(oc[v])
        print(f"{name:7s}{str(kw):32s} prec={'f32' if prec else 'f64'} T={len(oc):6d} K2={ms:8.3f}ms "
              f"maxrel={rel.max():.2e} med={np.median(rel):.2e} invalid_equal={np.array_equal(oc<0, gc<0)} "
              f"n_inv_diff={(np.not_equal(oc<0, gc<0)).sum()} best {ob['index']}=={gb['index']}")
