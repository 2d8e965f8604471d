#!/bin/bash
# Where the small-grid K1 (control cycles) spends its ~18 us at S = 40: ablation builds (-DSFW_K1S_ABL=1 no footprint tasks, 2 no
# sincos, 3 no atan2, 4 no serial velocity / heading recurrences) under rocprofv3 kernel stats of build/cycle_latency.
# Times only: the results of those builds are wrong by construction.
R=$(pwd); mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
cp $R/social_force_window_planner_amd/libsfw_hip.so /tmp/keep.so
for v in base 1 2 3 4; do
  if [ $v != base ]; then cp $R/build/libsfw_k1s$v.so $R/social_force_window_planner_amd/libsfw_hip.so; fi
  rm -rf /tmp/k1s_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/k1s_$v -- $R/build/cycle_latency 60 0 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys, statistics
d=[]
for f in glob.glob(f'/tmp/k1s_{sys.argv[1]}/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'rollout_small' in r['Kernel_Name']:
            d.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
# the tool runs 4 configurations with S = 40 first, then 4 with S = 6 (70 cycles each)
n=len(d)//2
print(f"{sys.argv[1]:>5}: S = 40 median {statistics.median(d[:n])/1e3:6.2f} us   S = 6 median {statistics.median(d[n:])/1e3:6.2f} us   ({len(d)} launches)")
PY
done
cp /tmp/keep.so $R/social_force_window_planner_amd/libsfw_hip.so
