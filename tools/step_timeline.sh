#!/bin/bash
# Device timeline of one blocking step of a workload (kernel + memory-copy trace): where the step's time outside K2 goes
WL=${1:-cfg2}
R=$(pwd); mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/stl -- python $R/bench.py --workload $WL --no-cpu-baseline --no-extra --no-verify --steps 12 --warmup 3 > /tmp/stl.out 2>&1
python3 - <<'PY'
import csv, glob, re
ev=[]
for f in glob.glob('/tmp/stl/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(sfw_\w+|__amd_\w+)', r['Kernel_Name'])
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), (m.group(1) if m else r['Kernel_Name'][:30]) + f" [{r['Grid_Size_X']}] q{r['Queue_Id']}"))
for f in glob.glob('/tmp/stl/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction'].replace('MEMORY_COPY_','')))
ev.sort()
k=[i for i,e in enumerate(ev) if 'sfw_argmin_stage1' in e[2]]
lo=k[-4]+1; hi=k[-2]+3   # two whole steps near the end
t0=ev[lo][0]; prev_end=None
for s,e,n in ev[lo:hi]:
    gap = '' if prev_end is None else f"(gap {(s-prev_end)/1e3:6.1f})"
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  {n:60s} {gap}")
    prev_end = max(prev_end or 0, e)
PY
