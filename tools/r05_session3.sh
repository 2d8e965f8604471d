#!/bin/bash
# Round 5, third GPU session: where a lone wave's laser-point pass spends its time (ablation builds: the evaluation loop
# without the reduction, the reduction without the loop) + the fixed test
mkdir -p gpurun_out
python -m pytest "tests/test_parity_holes_gpu.py" -x -q -m gpu -k "never_move" > gpurun_out/r05c_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05c_tests.log
NS=0,5,8,20,50; OS=0,16,60,240
{ echo "== default"; python tools/cycle_k2.py $NS $OS
  echo "== SFW_ABL_NOREDUCE"; SFW_HIP_LIB=build/libsfw_abl_noreduce.so python tools/cycle_k2.py $NS $OS
  echo "== SFW_ABL_NOOBSLOOP"; SFW_HIP_LIB=build/libsfw_abl_noobsloop.so python tools/cycle_k2.py $NS $OS
  echo "== default (again)"; python tools/cycle_k2.py $NS $OS; } > gpurun_out/r05_cycle_ablation.txt 2>&1
cat gpurun_out/r05_cycle_ablation.txt
