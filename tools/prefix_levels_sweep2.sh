for w in cfg2 target; do
  SFW_DEBUG_PLAN=1 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify 2>&1 | grep -m1 "shared prefix"
  for lv in "" "6,10" "6,12" "5,9,13" "6,10,14,18" "4,8,12,16" "7,14" "6,10,14,17,20" "3,6,9,12,15,18" "2,4,6,8,10,12,14,16,18,20" "4,8,12,16,20" "2,4,6,8,10,12,14,16,18"; do
    if [ -z "$lv" ]; then unset SFW_PREFIX; else export SFW_PREFIX=$lv; fi
    python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SFW_PREFIX=[$lv]', d['config']['workload'][:6], 'K2 %.4f ms' % d['kernel_ms']['social'], 'step %.4f ms' % d['ms_per_step'], 'clk %.2f' % d['sustained_clock_ghz'])"
  done
done
