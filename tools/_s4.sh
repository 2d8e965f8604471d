mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_cycle_kernel_gpu.py tests/test_multi_device_gpu.py -x -q -m gpu > gpurun_out/r06e_tests.log 2>&1; echo "tests rc=$?"; grep -n "passed\|failed" gpurun_out/r06e_tests.log; grep -B5 -A40 "^E " gpurun_out/r06e_tests.log | head -80
SFW_DEBUG_STAGE=1 python tools/host_gap_probe.py cfg2 30 2>&1 | grep "stage:" | tail -5
SFW_DEBUG_STAGE=1 python tools/host_gap_probe.py target 12 2>&1 | grep "stage:" | tail -3
SFW_DEBUG_STAGE=1 build/cycle_latency 30 0 2>&1 | grep "stage:" | sed -n '100,104p'
