#!/bin/bash
# Instruction counts per wave of the control-cycle K2 (lone waves) for several builds:  tools/pmc_cycle.sh "<N> <O>" lib1 lib2 ...  ("-" = in-tree)
ARGS=$1; shift
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$REPO/$lib; fi
  D=/tmp/pc_$$_$(basename $lib .so); rm -rf $D
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $D -- python $REPO/tools/cycle_k2.py $ARGS > /dev/null 2> $D.err
  echo "== $lib"
  python $REPO/tools/pmc_summary.py "$D/*/*counter_collection.csv" | grep -A12 "sfw_social" | grep -E "==|INSTS|WAVE_CYCLES|WAIT|WAVES"
done
