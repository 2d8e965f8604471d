#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_multi_device_gpu.py -x -q -m gpu > gpurun_out/s3_multi.log 2>&1; echo "multi rc=$?"; tail -12 gpurun_out/s3_multi.log | grep -v "^  File"
build/multi_device 2>&1 | tail -4
python -m pytest tests -x -q -m gpu --deselect tests/test_multi_device_gpu.py > gpurun_out/s3_gpu.log 2>&1; echo "gpu rc=$?"; tail -6 gpurun_out/s3_gpu.log | grep -v "^  File"
bash tools/profile.sh r02b_target target > gpurun_out/s3_prof_target.log 2>&1
bash tools/profile.sh r02b_cfg2 cfg2 > gpurun_out/s3_prof_cfg2.log 2>&1
grep -A 24 "sfw_social" gpurun_out/r02b_target_pmc_summary.txt | awk '/^==/{print} /> /{print}' | head -12
