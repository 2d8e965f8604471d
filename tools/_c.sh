TAG=r06m bash tools/session.sh tests tests/test_cycle_kernel_gpu.py tests/test_lifetime_gpu.py
python tools/cycle_k2.py 0,1,5,8,20,50 0,16,60,240 2>&1 | tail -8
build/cycle_latency 300 0
