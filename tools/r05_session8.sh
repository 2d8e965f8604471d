#!/bin/bash
# branch-free computeNewVelocity + parallel quotients in the small-grid K1: tests, kernel durations, cycle, cfg2 / target A/B
R=$(pwd); mkdir -p gpurun_out
python -m pytest tests/test_parity_gpu.py tests/test_golden.py tests/test_prefix_sharing_gpu.py tests/test_host_state_machine.py tests/test_parity_holes_gpu.py -x -q -m gpu > gpurun_out/r05i_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05i_tests.log
cd /tmp && export TMPDIR=/tmp
cp $R/social_force_window_planner_amd/libsfw_hip.so /tmp/keep.so
for v in new pre; do
  if [ $v = pre ]; then cp $R/build/libsfw_pre_k1b.so $R/social_force_window_planner_amd/libsfw_hip.so; fi
  rm -rf /tmp/k1b_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/k1b_$v -- $R/build/cycle_latency 60 0 > /dev/null 2>&1
  python3 - $v <<'PY'
import csv, glob, sys, statistics
d=[]
for f in glob.glob(f'/tmp/k1b_{sys.argv[1]}/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'rollout_small' in r['Kernel_Name']:
            d.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
n=len(d)//2
print(f"{sys.argv[1]:>5}: sfw_rollout_small_kernel S = 40 median {statistics.median(d[:n])/1e3:6.2f} us   S = 6 median {statistics.median(d[n:])/1e3:6.2f} us")
PY
done
for rep in 1 2; do for v in new pre; do
  if [ $v = pre ]; then cp $R/build/libsfw_pre_k1b.so $R/social_force_window_planner_amd/libsfw_hip.so; else cp /tmp/keep.so $R/social_force_window_planner_amd/libsfw_hip.so; fi
  echo "== $v ($rep)"; $R/build/cycle_latency 300 0 | grep "^N=" | cut -c1-40
done; done
cp /tmp/keep.so $R/social_force_window_planner_amd/libsfw_hip.so
cd $R
bash tools/ab_bench.sh build/libsfw_pre_k1b.so cfg2 target 2>&1 | sed 's/traj\/s//'
for v in - build/libsfw_pre_k1b.so; do
  if [ "$v" = "-" ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$R/$v; fi
  python bench.py --workload cfg2 --no-cpu-baseline --no-extra --no-verify 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'K1 %.4f ms' % d['kernel_ms']['rollout'], 'launch %.4f' % d['kernel_ms']['launch_total'])"
done
