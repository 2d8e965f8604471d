#!/bin/bash
# Timeline of the control cycle on the device: kernel + memory-copy trace of build/cycle_latency, gaps between the operations of a cycle
R=$(pwd); mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cyc
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/cyc -- $R/build/cycle_latency 40 0 > /tmp/cyc.out 2>&1
ls /tmp/cyc/*/
python3 - <<'PY' > $R/gpurun_out/cycle_timeline.txt
import csv, glob, re
ev=[]
for f in glob.glob('/tmp/cyc/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), (re.search(r'(sfw_\w+|__amd_\w+)', r['Kernel_Name']) or [r['Kernel_Name'][:40]])[0]))
for f in glob.glob('/tmp/cyc/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', ''))))
ev.sort()
# the second configuration the tool runs is N = 5, S = 40: the first launches that integrate pedestrians (> 30 us); skip 20
# cycles of it, print two cycles
k=[i for i,e in enumerate(ev) if 'sfw_social_kernel' in e[2] or 'sfw_cycle_kernel' in e[2]]
k=[i for i in k if ev[i][1]-ev[i][0] > 30000]   # (the robot-alone configuration's short launches come first)
mid=k[20]
lo=max(0,mid-5)
t0=ev[lo][0]
for s,e,n in ev[lo:lo+22]:
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:7.1f} us  {n}")
PY
cat $R/gpurun_out/cycle_timeline.txt
