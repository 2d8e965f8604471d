#!/bin/bash
# Device-only ISA of sfw_kernels.hip (~12 s) + per-loop instruction counts and register use of the two
# f64 K2 organisations:   tools/k2_isa.sh [extra hipcc flags]
OUT=${OUT:-/tmp/k2_isa.s}
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function --cuda-device-only -S "$@" \
  social_force_window_planner_amd/csrc/sfw_kernels.hip -o $OUT 2>&1 | grep -v "hip-link" 
for sym in sfw_social_kernel_flatIdLb0ELi64ELb0E sfw_social_kernel_flatIdLb0ELi64ELb1E sfw_social_kernelIdLi1ELb0E; do
  echo "== $sym"
  python tools/isa_loops.py $OUT $sym | awk '{ if ($6+0 >= 60) print }'
  awk -v s="$sym" '$0 ~ "^\t.set .*"s".*(num_vgpr|numbered_sgpr|private_seg_size)," {print "   " $2, $3}' $OUT
  grep -A40 "^\.amdhsa_kernel.*$sym" $OUT | grep "next_free_vgpr\|next_free_sgpr" | head -2
  L=$(grep -n "^_ZN12_GLOBAL__N_1[0-9]*$sym" $OUT | head -1 | cut -d: -f1); E=$(awk -v l=$L 'NR>l && /^\.Lfunc_end/ {print NR; exit}' $OUT)
  echo "   readlane in function: $(sed -n "${L},${E}p" $OUT | grep -c v_readlane)  scratch ops: $(sed -n "${L},${E}p" $OUT | grep -c 'scratch_\|buffer_store\|buffer_load')"
done
