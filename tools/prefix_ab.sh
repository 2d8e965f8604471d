#!/bin/bash
# shared-prefix rollout on/off inside one session
for rep in 1 2; do
  for pre in 0 -1; do
    for w in ${@:-cfg2 target}; do
      SFW_PREFIX=$pre python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SFW_PREFIX=$pre', d['config']['workload'][:6], '%.4g traj/s' % d['value'], 'K2 %.4f ms' % d['kernel_ms']['social'], 'step %.4f ms' % d['ms_per_step'], 'index', d['cmd_vel']['index'], 'cost', repr(d['cmd_vel']['cost']))"
    done
  done
done
