#!/bin/bash
# Round evidence in one GPU session: whole GPU suite, the default bench line, rocprofv3 stats + PMC passes of the
# headline workload and the two cfg2 variants.   tools/gpu_profile_round.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/${TAG}_gpu_tests.log
( time python bench.py ) > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench_default.err
bash tools/profile.sh ${TAG}_target target > gpurun_out/${TAG}_prof_target.log 2>&1
bash tools/profile.sh ${TAG}_cfg2 cfg2 > gpurun_out/${TAG}_prof_cfg2.log 2>&1
# cfg2 once more WITHOUT the split launch (build/libsfw_nosplit.so = -DSFW_SPLIT_FORMS=0): rocprofv3 serialises the split launch's two
# streams, so the kernel-stats of the shipped build do not add up to the HIP-event K2; this one's do
if [ -f build/libsfw_nosplit.so ]; then SFW_HIP_LIB=$(pwd)/build/libsfw_nosplit.so bash tools/profile.sh ${TAG}_cfg2_nosplit cfg2 > gpurun_out/${TAG}_prof_cfg2_nosplit.log 2>&1; fi
bash tools/profile.sh ${TAG}_cfg2_o64 cfg2_o64 > gpurun_out/${TAG}_prof_cfg2_o64.log 2>&1
bash tools/profile.sh ${TAG}_target_o720 target_o720 > gpurun_out/${TAG}_prof_target_o720.log 2>&1
{ for o in 0 60 240; do echo "== build/cycle_latency 300 $o"; build/cycle_latency 300 $o; done; echo "== build/cycle_latency 300 0 1 (marker capture)"; build/cycle_latency 300 0 1; } > gpurun_out/${TAG}_latency.txt 2>&1
python tools/traffic_json.py target=gpurun_out/${TAG}_target_traffic.txt cfg2=gpurun_out/${TAG}_cfg2_traffic.txt cfg2_o64=gpurun_out/${TAG}_cfg2_o64_traffic.txt target_o720=gpurun_out/${TAG}_target_o720_traffic.txt > gpurun_out/${TAG}_traffic.json
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_traffic.json'))
for k,v in d.items(): print(k, 'K2 MB %.1f' % (v['k2_bytes']/1e6), 'all MB %.1f' % (v['all_kernels_bytes']/1e6))"
