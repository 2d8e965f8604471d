#!/bin/bash
# host enqueue order: levels first, side-stream footprint work after them; the timed launch's clock-probe memset in front of the rollout
mkdir -p gpurun_out
python -m pytest tests/test_lifetime_gpu.py tests/test_abi_errors_gpu.py tests/test_prefix_sharing_gpu.py tests/test_parity_gpu.py tests/test_multi_device_gpu.py tests/test_handle_fuzz_gpu.py -x -q -m gpu > gpurun_out/r05j_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r05j_tests.log
bash tools/ab_bench.sh build/libsfw_pre_order.so cfg2 cfg2_o64 target cfg3 2>&1 | sed 's/traj\/s//'
