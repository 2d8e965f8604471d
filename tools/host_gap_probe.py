#!/usr/bin/env python3
"""Where the host's share of a blocking step goes (stage / launch / fetch wall-clock per call, the loop's own overhead):
   python tools/host_gap_probe.py [workload] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
job = bench.GridJob(name, "f64", 0, 1, 0)
s, sc = job.scorer, job.scene
buf = np.empty(job.n_local)
for _ in range(5):
    job.step()
t = {"stage": [], "launch": [], "fetch": [], "step": []}
for _ in range(steps):
    t0 = time.perf_counter()
    s.stage(sc.robot_state, job.lin, job.ang, sc.goal_args, job.index_base)
    t1 = time.perf_counter()
    s.launch()
    t2 = time.perf_counter()
    s.fetch(want_costs=True, out=buf)
    t3 = time.perf_counter()
    t["stage"].append(t1 - t0); t["launch"].append(t2 - t1); t["fetch"].append(t3 - t2); t["step"].append(t3 - t0)
print(name, " ".join(f"{k} {np.median(v) * 1e6:.1f} us" for k, v in t.items()))
# the same without the cost vector (what the fetch's memcpy of the vector costs)
f2 = []
for _ in range(steps):
    s.stage(sc.robot_state, job.lin, job.ang, sc.goal_args, job.index_base); s.launch()
    t2 = time.perf_counter(); s.fetch(want_costs=False); f2.append(time.perf_counter() - t2)
print(name, f"fetch without the vector {np.median(f2) * 1e6:.1f} us")
