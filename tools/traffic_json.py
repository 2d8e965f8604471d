"""Build profiles/r01_traffic.json from the FETCH_SIZE/WRITE_SIZE summaries written by tools/profile.sh.
usage: traffic_json.py <workload>=<traffic.txt>[:<K2 launches per step>] ... > profiles/r01_traffic.json
(the summaries hold means per dispatch; with the shared-prefix rollout K2 is two dispatches per step)"""
import json
import re
import sys

out = {}
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    per_step = 1
    if ":" in path:
        path, n = path.rsplit(":", 1)
        per_step = int(n)
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"== (\S.*?)\s+dispatches=", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)", line)
        if m and cur is not None:
            cur["fetch_bytes" if m.group(1) == "FETCH_SIZE" else "write_bytes"] = float(m.group(2)) * 1024.0
    kernels = {k: v for k, v in kernels.items() if "pair_table" not in k}
    k2 = next(k for k in kernels if "social" in k)
    out[name] = {
        "kernels": kernels,
        "k2": k2,
        "k2_launches_per_step": per_step,
        "k2_bytes": per_step * (kernels[k2]["fetch_bytes"] + kernels[k2]["write_bytes"]),
        "all_kernels_bytes": sum((per_step if k == k2 else 1) * (v.get("fetch_bytes", 0) + v.get("write_bytes", 0))
                                 for k, v in kernels.items()),
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile.sh), KB x 1024, "
                  "mean per dispatch; 32-byte-per-lane records, so the gfx950 half-reporting of 16 B/lane streams "
                  "does not apply (K2 fetch == K1->K2 robot-step table size)",
    }
json.dump(out, sys.stdout, indent=1)
