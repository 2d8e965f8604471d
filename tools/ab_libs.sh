#!/bin/bash
# K2 time / sustained clock of several builds of libsfw_hip.so on several workloads inside ONE gpurun call:
#   tools/ab_libs.sh "<wl1> <wl2> ..." lib1.so lib2.so ...     ("-" = the in-tree build); three alternations
WLS=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$PWD/$lib; fi
    for w in $WLS; do
      python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$lib', '%-12s' % d['config']['workload'].split(':')[0], '%.4g traj/s' % d['value'], 'K2 %.4f ms' % d['kernel_ms']['social'], 'clock %.3f GHz' % d['sustained_clock_ghz'], 'frac %.3f' % d['roofline']['frac'])"
    done
  done
done
