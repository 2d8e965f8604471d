for w in cfg2 target; do
  for pre in -1 4 6 8 10 12 14 16; do
    SFW_DEBUG_PLAN=1 SFW_PREFIX=$pre python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify 2>/tmp/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SFW_PREFIX=$pre', d['config']['workload'][:6], 'K2 %.4f ms' % d['kernel_ms']['social'], 'step %.4f ms' % d['ms_per_step'])"
    grep -m1 "shared prefix" /tmp/err.txt
  done
done
