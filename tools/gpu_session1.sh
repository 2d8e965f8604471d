#!/bin/bash
# round-2 first GPU session: new parity tests, whole GPU suite, default bench line, profiles
mkdir -p gpurun_out
python -m pytest tests/test_parity_holes_gpu.py -x -q -m gpu > gpurun_out/s1_holes.log 2>&1; echo "holes rc=$?"; tail -15 gpurun_out/s1_holes.log
python -m pytest tests -x -q -m gpu --deselect tests/test_parity_holes_gpu.py > gpurun_out/s1_gpu.log 2>&1; echo "gpu rc=$?"; tail -5 gpurun_out/s1_gpu.log
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/s1_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/s1_smoke.log
( time python bench.py ) > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/s1_bench.err
bash tools/profile.sh r02a_target target > gpurun_out/s1_prof_target.log 2>&1
bash tools/profile.sh r02a_cfg2 cfg2 > gpurun_out/s1_prof_cfg2.log 2>&1
