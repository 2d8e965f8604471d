for w in cfg2 target cfg3; do SFW_DEBUG_PLAN=1 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify 2>&1 | grep -m1 "shared prefix"; done
for lv in "10" "4,10" "3,6,10" "2,4,6,8,10,12" "1,2,3,4,5,6,7,8,9,10,11,12,13,14" "3,6,9,12,15" "4,8,12,16"; do
  for w in cfg2 target; do
    SFW_PREFIX=$lv python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SFW_PREFIX=$lv', d['config']['workload'][:6], 'K2 %.4f ms' % d['kernel_ms']['social'], 'step %.4f ms' % d['ms_per_step'])"
  done
done
