#!/bin/bash
# A/B two builds of libsfw_hip.so inside ONE gpurun call (boxes differ by ~10 % in sustained clock):
#   tools/ab_bench.sh <libB.so> [workloads...]   — alternates A (in-tree) and B three times
B=$1; shift
WLS=${@:-cfg2 target}
for rep in 1 2 3; do
  for v in A B; do
    for w in $WLS; do
      if [ $v = A ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$B; fi
      python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'][:6], '%.4g traj/s' % d['value'], 'K2 %.4f ms' % d['kernel_ms']['social'], 'step %.4f ms' % d['ms_per_step'])"
    done
  done
done
