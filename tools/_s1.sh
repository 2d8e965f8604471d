mkdir -p gpurun_out
python -m pytest tests/test_prefix_sharing_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r06b_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06b_tests.log
ab() { for rep in 1 2 3; do for lib in build/libsfw_r05.so -; do
  if [ "$lib" = "-" ]; then unset SFW_HIP_LIB; else export SFW_HIP_LIB=$PWD/$lib; fi
  for w in cfg2 target; do python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-22s' % '$lib', '%-8s' % d['config']['workload'].split(':')[0], 'step %.4f ms' % d['ms_per_step'], 'median %.4f' % d['median_ms_per_step'], 'K1 %.4f K2 %.4f' % (d['kernel_ms']['rollout'], d['kernel_ms']['social']), 'clock %.3f' % d['sustained_clock_ghz'])"; done; done; done; unset SFW_HIP_LIB; }
ab | tee gpurun_out/r06b_ab_head.txt
bash tools/step_timeline.sh cfg2 > gpurun_out/r06b_step_timeline_cfg2.txt 2>&1; cat gpurun_out/r06b_step_timeline_cfg2.txt
