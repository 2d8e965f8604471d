"""Kernel resource table from `make -C csrc resources` output on stdin."""
import re
import sys

rows, cur = [], None
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1).replace("_ZN12_GLOBAL__N_1", "")[2:50]}
        rows.append(cur)
    for key, pat in (("vgpr", r"remark:\s+VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                     ("sspill", r"SGPRs Spill: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r"remark:\s+SGPRs: (\d+)")):
        m = re.search(pat, l)
        if m and cur is not None:
            cur[key] = m.group(1)
print(f"{'kernel':50s} vgpr sgpr occ vspill sspill")
for r in rows:
    print(f"{r['name']:50s} {r.get('vgpr','?'):>4} {r.get('sgpr','?'):>4} {r.get('occ','?'):>3} {r.get('vspill','?'):>6} {r.get('sspill','?'):>6}")
