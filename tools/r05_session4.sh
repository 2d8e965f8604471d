#!/bin/bash
# Round 5, fourth GPU session: the two-level summation order of the laser-point sums + the wide LDS phase of a lone wave:
# bit-identity of the forms, parity, the control-cycle grid against the build before it and against round 3's kernels, and
# what the GPU-filling launches pay or gain.
mkdir -p gpurun_out
python -m pytest tests/test_k2_forms_gpu.py tests/test_parity_holes_gpu.py tests/test_parity_gpu.py tests/test_random_scenes_gpu.py -x -q -m gpu > gpurun_out/r05f_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05f_tests.log
python tools/kernel_equiv.py > gpurun_out/r05f_kernel_equiv.txt 2>&1; tail -3 gpurun_out/r05f_kernel_equiv.txt
NS=0,1,5,8,12,20,30,50; OS=16,60,120,240,720
{ echo "== round 5, wide LDS phase (this build)"; python tools/cycle_k2.py $NS $OS
  echo "== build/libsfw_pre_reduce.so (the build before it)"; SFW_HIP_LIB=build/libsfw_pre_reduce.so python tools/cycle_k2.py $NS $OS
  echo "== build/libsfw_hip_soz1.so (round 3)"; SFW_HIP_LIB=build/libsfw_hip_soz1.so python tools/cycle_k2.py $NS $OS
  echo "== SFW_OBS_TASKS=0 (this build, one lane per agent)"; SFW_OBS_TASKS=0 python tools/cycle_k2.py $NS $OS
  echo "== round 5, wide LDS phase (again)"; python tools/cycle_k2.py $NS $OS; } > gpurun_out/r05_cycle_k2.txt 2>&1
tail -9 gpurun_out/r05_cycle_k2.txt
bash tools/ab_bench.sh build/libsfw_pre_reduce.so cfg2_o64 cfg2_o240 target_o720 > gpurun_out/r05_ab_reduce.txt 2>&1; cat gpurun_out/r05_ab_reduce.txt
