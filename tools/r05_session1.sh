#!/bin/bash
# Round 5, first GPU session: the GPU suite on the degree-9 default, the two 3000-seed sweeps, the same-session price of the
# degree (against build/libsfw_e8.so), the cfg2 row-block probe, one full default bench line.
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -rA > gpurun_out/r05a_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r05a_gpu_tests.log | tail -2
grep "chaos allowance" gpurun_out/r05a_gpu_tests.log | tail -2
python tools/sweep_parity.py 100 3000 > gpurun_out/r05_sweep_a.txt 2>&1; tail -1 gpurun_out/r05_sweep_a.txt
python tools/sweep_parity.py 20000 3000 > gpurun_out/r05_sweep_b.txt 2>&1; tail -1 gpurun_out/r05_sweep_b.txt
bash tools/ab_bench.sh build/libsfw_e8.so cfg2 target > gpurun_out/r05_ab_exp9.txt 2>&1; cat gpurun_out/r05_ab_exp9.txt
python bench.py --workload cfg2 --extras inproc_multi --no-cpu-baseline > gpurun_out/r05_cfg2_inproc.json 2> gpurun_out/r05_cfg2_inproc.err; echo "cfg2 inproc rc=$?"
( time python bench.py ) > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench rc=$?"; tail -4 gpurun_out/r05a_bench.err
