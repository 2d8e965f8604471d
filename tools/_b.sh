mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r06k_bench_default.json 2> gpurun_out/r06k_bench_default.err; echo "bench rc=$?"; tail -5 gpurun_out/r06k_bench_default.err
python tools/bench_table.py gpurun_out/r06k_bench_default.json 2>&1 | head -40
TAG=r06k bash tools/session.sh cycle 300 | head -40
python -c "import __graft_entry__ as g; g.smoke()"
