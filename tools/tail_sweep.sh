#!/bin/bash
# K2 per sample over grids of 8 ... 10.7 register-form wave layers (cfg2 crowd, 21 agents), automatic organisation and forced flat:
# is there a tail when the waves of a register-form launch do not divide evenly over the SIMDs and queue (more than six layers)?
for g in 192x128 196x128 200x128 208x128 224x128 240x128 256x128; do
  for form in auto flat; do
    if [ $form = flat ]; then export SFW_FORCE_FLAT=1; else unset SFW_FORCE_FLAT; fi
    python bench.py --workload cfg2 --grid $g --no-cpu-baseline --no-extra --no-verify --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$g $form', '%.4g traj/s' % d['value'], 'K2 %.4f ms' % d['kernel_ms']['social'], 'per-sample-ns %.2f' % (d['kernel_ms']['social']*1e6/ (int('$g'.split('x')[0])*128)), 'clock %.2f' % d['sustained_clock_ghz'])"
  done
done
