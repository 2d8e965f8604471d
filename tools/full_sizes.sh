#!/bin/bash
# The large BASELINE configs at full size on one GPU (cfg5: one rank's shard of 8).
for spec in "cfg3" "cfg4" "cfg5 --grid 512x4096"; do
  python bench.py --workload $spec --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-verify 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$spec', '%.4g traj/s' % d['value'], '%.2f ms/step' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms'].items()}, 'frac %.3f' % d['roofline']['frac'], 'n_valid', d['cmd_vel']['n_valid'], 'index', d['cmd_vel']['index'])"
done
