#!/bin/bash
# rocprofv3 kernel stats of the control-cycle microbenchmark (examples/cycle_latency.cpp, 8 configurations x 210 cycles):
#   tools/latency_profile.sh [laser points] [tag]  -> gpurun_out/<tag>_latency_kernel_stats.csv, gpurun_out/<tag>_latency.txt
O=${1:-0}
TAG=${2:-r03}
R=$(pwd)
mkdir -p $R/gpurun_out
$R/build/cycle_latency 300 $O | grep "^N=" > $R/gpurun_out/${TAG}_latency.txt
$R/build/cycle_latency 300 60 | grep "^N=" >> $R/gpurun_out/${TAG}_latency.txt
$R/build/cycle_latency 300 240 | grep "^N=" >> $R/gpurun_out/${TAG}_latency.txt
echo "# with the Trajectory points of all 45 samples fetched every cycle (sfw_set_points_capture)" >> $R/gpurun_out/${TAG}_latency.txt
$R/build/cycle_latency 300 0 1 | grep "^N=" >> $R/gpurun_out/${TAG}_latency.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/latprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/latprof -- $R/build/cycle_latency 200 $O > /dev/null 2>&1
cp $(ls /tmp/latprof/*/*kernel_stats.csv | head -1) $R/gpurun_out/${TAG}_latency_kernel_stats.csv
cat $R/gpurun_out/${TAG}_latency_kernel_stats.csv | cut -d, -f1-4
