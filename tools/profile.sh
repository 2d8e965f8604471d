#!/bin/bash
# Collect the rocprofv3 evidence for one workload on the GPU box:
#   tools/profile.sh <tag> [workload]     -> gpurun_out/<tag>_{kernel_stats.csv,bench.json,pmc_summary.txt,traffic.txt}
# Kernel trace + stats in one run; every PMC group in its own run with --kernel-trace only.
set -u
TAG=$1
WL=${2:-cfg2}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload $WL --no-cpu-baseline --no-extra --no-verify"
D=/tmp/prof_$TAG
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $BENCH > $OUT/${TAG}_bench.json 2> $D.err
cp $(ls $D/stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats.csv
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $D/pmc$i -- $BENCH --steps 5 --warmup 1 > /dev/null 2>> $D.err
done
python $REPO/tools/pmc_summary.py "$D/pmc1/*/*counter_collection.csv" "$D/pmc2/*/*counter_collection.csv" > $OUT/${TAG}_pmc_summary.txt
python $REPO/tools/pmc_summary.py "$D/pmc3/*/*counter_collection.csv" "$D/pmc4/*/*counter_collection.csv" > $OUT/${TAG}_traffic.txt
tail -1 $OUT/${TAG}_bench.json | cut -c1-300
head -12 $OUT/${TAG}_kernel_stats.csv
cat $OUT/${TAG}_pmc_summary.txt | head -60
