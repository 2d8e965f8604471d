#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per DISPATCH KIND.

A bench step dispatches the social kernel several times (one launch per shared-prefix level, then the
suffix launch over all samples); these are different amounts of work, so they are kept apart: a kind is
(kernel, grid size).  Derived ratios the profiles/README quotes are appended per kind when their counters
are present.
"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in sys.argv[1:]:
    for f in glob.glob(path):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            kind = (k, int(r.get("Grid_Size", 0) or 0))
            acc[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[kind].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for kind in sorted(acc, key=lambda q: (q[0], -q[1])):
    k, grid = kind
    if "sfw" not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in acc[kind].items()}
    print(f"== {k}  grid={grid} ({grid // 64} waves)  dispatches={len(dur[kind])} mean_ns={sum(dur[kind]) / len(dur[kind]):.0f}")
    for c, v in sorted(m.items()):
        print(f"   {c:28s} {v:16.1f}")
    if m.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   {'> LDS bank conflict share':28s} {m.get('SQ_LDS_BANK_CONFLICT', 0.0) / m['SQ_LDS_IDX_ACTIVE']:16.4f}"
              "   (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)")
    if m.get("SQ_WAVES") and m.get("SQ_INSTS_VALU"):
        print(f"   {'> VALU insts per wave':28s} {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:16.1f}")
    if m.get("SQ_ACTIVE_INST_VALU") and m.get("SQ_BUSY_CYCLES"):
        # SQ_ACTIVE_INST_VALU counts per-SIMD cycles /4 summed over SEs; see profiles/README.md for the normalisation
        print(f"   {'> ACTIVE_INST_VALU/BUSY_CYC':28s} {m['SQ_ACTIVE_INST_VALU'] / m['SQ_BUSY_CYCLES']:16.4f}")
