#!/bin/bash
mkdir -p gpurun_out
python tools/kernel_equiv.py > gpurun_out/s2_equiv.log 2>&1; echo "equiv rc=$?"; cat gpurun_out/s2_equiv.log | tail -12
python -m pytest tests -x -q -m gpu > gpurun_out/s2_gpu.log 2>&1; echo "gpu rc=$?"; tail -8 gpurun_out/s2_gpu.log
bash tools/ab_bench.sh $PWD/build/libsfw_hip_base.so target cfg2 cfg2_o64 > gpurun_out/s2_ab.log 2>&1; cat gpurun_out/s2_ab.log
