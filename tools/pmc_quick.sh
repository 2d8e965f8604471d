#!/bin/bash
# VALU / LDS utilisation of one workload in two short PMC passes:  tools/pmc_quick.sh <tag> <workload> [bench args]
TAG=$1; WL=$2; shift 2
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D=/tmp/pq_$TAG; rm -rf $D
BENCH="python $REPO/bench.py --workload $WL --no-cpu-baseline --no-extra --no-verify --steps 2 --warmup 1 $@"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $D/p1 -- $BENCH > /dev/null 2> $D.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $D/p2 -- $BENCH > /dev/null 2>> $D.err
python $REPO/tools/pmc_summary.py "$D/p1/*/*counter_collection.csv" "$D/p2/*/*counter_collection.csv" > $OUT/${TAG}_pmc_summary.txt
grep -n "sfw_social" -A19 $OUT/${TAG}_pmc_summary.txt | grep -E "==|GRBM|ACTIVE_INST_VALU|LDS_IDX_ACTIVE|INSTS_VALU|WAVES |BANK" 
