#!/bin/bash
mkdir -p gpurun_out
python tools/kernel_equiv.py > gpurun_out/s5_equiv.log 2>&1; echo "equiv rc=$?"; grep -c "flat==reg True  default==flat True" gpurun_out/s5_equiv.log; grep -v "flat==reg True  default==flat True" gpurun_out/s5_equiv.log | tail -5
python -m pytest tests -x -q -m gpu > gpurun_out/s5_gpu.log 2>&1; echo "gpu rc=$?"; tail -4 gpurun_out/s5_gpu.log | grep -v "^  File"
for o in 60 240; do
echo "--- new O=$o"; build/cycle_latency 300 $o 2>&1 | grep "^N="
done
bash tools/ab_bench.sh $PWD/build/libsfw_hip_prev.so target cfg2 cfg2_o64 2>&1 | grep "^[AB] "
