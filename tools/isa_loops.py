"""Count VALU / SALU / LDS / VMEM instructions per loop (backward branch) of one kernel in a
hipcc -save-temps .s listing.  usage: isa_loops.py <file.s> <symbol substring>"""
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(n for n, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and sym in l.split(":")[0])
    end = next(n for n in range(start, len(lines)) if lines[n].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    for n, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = n
    for n, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            seg = body[labels[m.group(1)]:n + 1]
            cnt = lambda p: sum(1 for x in seg if re.match(r"\s+" + p, x))
            f64 = sum(1 for x in seg if re.match(r"\s+v_\w+_f64", x))
            trans = sum(1 for x in seg if re.match(r"\s+v_(rcp|rsq|sqrt|exp|log|sin|cos)_", x))
            print(f"{m.group(1):>12} lines {labels[m.group(1)]:5d}-{n:5d}  valu {cnt('v_'):4d} (f64 {f64}, trans {trans})"
                  f"  salu {cnt('s_'):4d}  ds {cnt('ds_'):3d}  vmem {cnt('(global|buffer|flat|scratch)_'):3d}")


main()
