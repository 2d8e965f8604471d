#!/bin/bash
mkdir -p gpurun_out
python tools/kernel_equiv.py > gpurun_out/s4_equiv.log 2>&1; echo "equiv rc=$?"; grep -c "flat==reg True  default==flat True" gpurun_out/s4_equiv.log; grep -v "flat==reg True  default==flat True" gpurun_out/s4_equiv.log | tail -5
python -m pytest tests -x -q -m gpu > gpurun_out/s4_gpu.log 2>&1; echo "gpu rc=$?"; tail -6 gpurun_out/s4_gpu.log | grep -v "^  File"
echo "--- latency new (N=5, S=40): O=0,60,240"; for o in 0 60 240; do build/cycle_latency 300 $o 2>&1 | tail -1; done
echo "--- latency prev"; for o in 0 60 240; do LD_LIBRARY_PATH=$PWD/build/prev build/cycle_latency 300 $o 2>&1 | tail -1; done
bash tools/ab_bench.sh $PWD/build/libsfw_hip_prev.so target cfg2 cfg2_o64 2>&1 | grep "^[AB] "
