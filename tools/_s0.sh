mkdir -p gpurun_out
python -m pytest tests/test_bench_gpu.py -x -q -m gpu -k "two_ranks or single_gpu" > gpurun_out/r06a_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r06a_tests.log
bash tools/step_timeline.sh cfg2 > gpurun_out/r06a_step_timeline_cfg2.txt 2>&1; cat gpurun_out/r06a_step_timeline_cfg2.txt
bash tools/step_timeline.sh target > gpurun_out/r06a_step_timeline_target.txt 2>&1; head -30 gpurun_out/r06a_step_timeline_target.txt
build/cycle_latency 300 0 | tee gpurun_out/r06a_latency.txt
for w in cfg2 target; do python bench.py --workload $w --no-cpu-baseline --no-extra --no-verify --steps 20 --warmup 3 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:10], d['value'], d['ms_per_step'], d['median_ms_per_step'], d['kernel_ms'])"; done
