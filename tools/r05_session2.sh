#!/bin/bash
# Round 5, second GPU session: the new / changed tests, the control-cycle grid with the laser-point pass forced either way
# (item 5), the crowd-size quantisation of the task loop (item 6).
mkdir -p gpurun_out
python -m pytest tests/test_parity_holes_gpu.py tests/test_bench_gpu.py::test_single_gpu_line -x -q -m gpu > gpurun_out/r05b_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05b_tests.log
NS=0,1,2,3,5,8,12,16,20,30,50; OS=16,60,120,240,720
{ echo "== auto"; python tools/cycle_k2.py $NS $OS
  echo "== SFW_OBS_TASKS=1"; SFW_OBS_TASKS=1 python tools/cycle_k2.py $NS $OS
  echo "== SFW_OBS_TASKS=0"; SFW_OBS_TASKS=0 python tools/cycle_k2.py $NS $OS
  echo "== build/libsfw_hip_soz1.so (round 3)"; SFW_HIP_LIB=build/libsfw_hip_soz1.so python tools/cycle_k2.py $NS $OS
  echo "== auto (again)"; python tools/cycle_k2.py $NS $OS; } > gpurun_out/r05_cycle_forms.txt 2>&1
tail -14 gpurun_out/r05_cycle_forms.txt
python tools/crowd_quantisation.py > gpurun_out/r05_crowd_quantisation.txt 2>&1; cat gpurun_out/r05_crowd_quantisation.txt
