#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel."""
import csv, sys, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in sys.argv[1:]:
    for f in glob.glob(path):
        seen=set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key=(f,r["Dispatch_Id"])
            if key not in seen:
                seen.add(key); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k in acc:
    if "sfw" not in k: continue
    print(f"== {k}  dispatches={len(dur[k])} mean_ns={sum(dur[k])/len(dur[k]):.0f}")
    for c,v in sorted(acc[k].items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}")
