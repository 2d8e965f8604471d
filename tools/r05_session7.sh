#!/bin/bash
# the small-grid K1 (control cycles): chunked serial loops + grouped cell loads, against the build before it
mkdir -p gpurun_out
python -m pytest tests/test_parity_gpu.py tests/test_host_state_machine.py tests/test_golden.py tests/test_lifetime_gpu.py -x -q -m gpu > gpurun_out/r05h_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05h_tests.log
python tools/kernel_equiv.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for lib in new pre; do
  rm -rf /tmp/lp_$lib
  if [ $lib = pre ]; then export LD_PRELOAD=; cp $GRAFT_REPO_ROOT/social_force_window_planner_amd/libsfw_hip.so /tmp/new.so; cp $GRAFT_REPO_ROOT/build/libsfw_pre_k1small.so $GRAFT_REPO_ROOT/social_force_window_planner_amd/libsfw_hip.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$lib -- $GRAFT_REPO_ROOT/build/cycle_latency 200 0 > /tmp/lat_$lib.txt 2>&1
  echo "== $lib"; grep "^N=" /tmp/lat_$lib.txt | head -8
  grep -E "rollout_small|argmin|social_kernel" $(ls /tmp/lp_$lib/*/*kernel_stats.csv | head -1) | cut -d, -f1-4
done
cp /tmp/new.so $GRAFT_REPO_ROOT/social_force_window_planner_amd/libsfw_hip.so
for rep in 1 2; do for lib in new pre; do
  if [ $lib = pre ]; then cp $GRAFT_REPO_ROOT/build/libsfw_pre_k1small.so $GRAFT_REPO_ROOT/social_force_window_planner_amd/libsfw_hip.so; else cp /tmp/new.so $GRAFT_REPO_ROOT/social_force_window_planner_amd/libsfw_hip.so; fi
  echo "== $lib (no profiler, $rep)"; $GRAFT_REPO_ROOT/build/cycle_latency 300 0 | grep "^N="
done; done
cp /tmp/new.so $GRAFT_REPO_ROOT/social_force_window_planner_amd/libsfw_hip.so
