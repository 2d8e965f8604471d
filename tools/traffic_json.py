"""Build profiles/r0N_traffic.json from the FETCH_SIZE/WRITE_SIZE summaries written by tools/profile.sh.
usage: traffic_json.py <workload>=<traffic.txt> ... > profiles/r02_traffic.json
The summaries hold means per dispatch; a bench step dispatches K2 several times (shared-prefix levels +
suffix, possibly in both kernel organisations), so bytes are converted to per-step sums using the
dispatch counts (one sfw_costmap_scan_kernel dispatch per step; sfw_rollout_kernel also runs in stage-only calls, e.g.
bench.py's plan query, since the stage enqueues it)."""
import json
import re
import sys

out = {}
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"== (\S.*?)\s+dispatches=(\d+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {"dispatches": int(m.group(2))})
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)", line)
        if m and cur is not None:
            cur["fetch_bytes" if m.group(1) == "FETCH_SIZE" else "write_bytes"] = float(m.group(2)) * 1024.0
    kernels = {k: v for k, v in kernels.items() if "pair_table" not in k}
    per_step_kernel = "sfw_costmap_scan_kernel" if any(k.startswith("sfw_costmap_scan_kernel") for k in kernels) else "sfw_rollout_kernel"
    steps = next(v["dispatches"] for k, v in kernels.items() if k.startswith(per_step_kernel))
    per_step = {}
    for k, v in kernels.items():
        n = v["dispatches"] / steps
        if k.startswith("sfw_rollout_kernel"):
            n = 1.0  # once per step; the surplus dispatches are stage-only calls outside any step
        per_step[k] = {"dispatches_per_step": n, "fetch_bytes": v.get("fetch_bytes", 0.0) * n,
                       "write_bytes": v.get("write_bytes", 0.0) * n}
    k2 = [k for k in per_step if "social" in k]
    out[name] = {
        "kernels": per_step,
        "k2": k2,
        "k2_bytes": sum(per_step[k]["fetch_bytes"] + per_step[k]["write_bytes"] for k in k2),
        "all_kernels_bytes": sum(v["fetch_bytes"] + v["write_bytes"] for v in per_step.values()),
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile.sh), KB x 1024, "
                  "mean per dispatch x dispatches per bench step; 32-byte-per-lane records, so the gfx950 "
                  "half-reporting of 16 B/lane streams does not apply",
    }
json.dump(out, sys.stdout, indent=1)
