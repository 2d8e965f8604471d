#!/bin/bash
# Times several builds of libsfw_hip.so in one GPU session (ablation / variant studies): tools/abl_bench.sh "<workloads>" lib1.so lib2.so ...
# ("-" = the in-tree build).  Results of ablated kernels are wrong by construction; only the times mean something.
WLS=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    for w in $WLS; do
      if [ "$lib" = "-" ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$lib; fi
      python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s' % '$lib', d['config']['workload'][:8], 'K2 %.4f ms' % d['kernel_ms']['social'], 'step %.4f ms' % d['ms_per_step'], 'clk %.2f' % (d.get('sustained_clock_ghz') or 0))"
    done
  done
done
