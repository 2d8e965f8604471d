#!/bin/bash
# A/B: the flat form's first 64 pair-table entries in registers across the rollout (this build) against the build before it
mkdir -p gpurun_out
python -m pytest tests/test_k2_forms_gpu.py tests/test_prefix_sharing_gpu.py -x -q -m gpu > gpurun_out/r05g_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05g_tests.log
NS=1,5,8,20,50; OS=0,60,240
{ for rep in 1 2; do
  echo "== this build ($rep)"; python tools/cycle_k2.py $NS $OS
  echo "== build/libsfw_pre_first.so ($rep)"; SFW_HIP_LIB=build/libsfw_pre_first.so python tools/cycle_k2.py $NS $OS
done; } > gpurun_out/r05_cycle_first_pairs.txt 2>&1
cat gpurun_out/r05_cycle_first_pairs.txt
bash tools/ab_bench.sh build/libsfw_pre_first.so cfg2 target > gpurun_out/r05_ab_first_pairs.txt 2>&1; cat gpurun_out/r05_ab_first_pairs.txt
