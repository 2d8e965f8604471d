#!/bin/bash
# Experiment: the item count up to which a launch takes the flat form (16 per CU = 4096 items) raised to 24 / 32 per CU — cfg2's
# 6072-item prefix level (2024 register-form waves: two per SIMD, 43 % VALU-active) would run as 6072 flat waves
mkdir -p gpurun_out
{ bash tools/ab_libs.sh "cfg2 cfg2_o64" - build/libsfw_flat24.so build/libsfw_flat32.so
  for lib in - build/libsfw_flat24.so build/libsfw_flat32.so; do
    if [ "$lib" = "-" ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$PWD/$lib; fi
    for g in 72x72 80x80 96x96; do
      python bench.py --workload cfg2 --grid $g --no-cpu-baseline --no-extra --no-verify --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$lib', '$g', '%.4g traj/s' % d['value'], 'K2 %.4f ms' % d['kernel_ms']['social'])"
    done
  done; } > gpurun_out/r05_flat_threshold.txt 2>&1
cat gpurun_out/r05_flat_threshold.txt
