#!/bin/bash
# What a GPU session of this repository runs, as ONE parameterised script (gpurun -- 'bash tools/session.sh <what> ...');
# everything lands under gpurun_out/ with the tag TAG (environment, default r06).
#   tests [pytest args]          the -m gpu suite (default: all of tests/)            -> <TAG>_tests.log
#   ab "<workloads>" [A] [B]...  same-session A/B of whole builds, three alternations: an argument is a libsfw_hip.so
#                                (SFW_HIP_LIB), a directory holding another checkout of bench.py + package (e.g.
#                                build/r05_tree: `git archive <rev> bench.py social_force_window_planner_amd oracle | tar -x
#                                -C build/r05_tree` + its built library), or "-" = this tree                -> <TAG>_ab.txt
#   timeline <workload>          device timeline of two blocking steps (tools/step_timeline.sh)  -> <TAG>_step_timeline_<w>.txt
#   cycle [cycles]               control-cycle latency (build/cycle_latency, 0 / 60 / 240 points, marker capture) + its device
#                                timeline, the one-launch kernel against SFW_CYCLE_FUSED=0      -> <TAG>_latency.txt, <TAG>_cycle_timeline.txt
#   hostgap <workload>           host share of a blocking step (tools/host_gap_probe.py, SFW_DEBUG_STAGE)
#   evidence                     the round's evidence set, ONCE, on the final commit: suite, default bench line, rocprofv3 stats
#                                + PMC passes (tools/gpu_profile_round.sh), cycle, timelines, the two parity sweeps, the
#                                control-cycle K2 grid
TAG=${TAG:-r06}
OUT=gpurun_out
mkdir -p $OUT
what=$1; shift
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-24s' % '$1', '%-10s' % d['config']['workload'].split(':')[0], '%.4g traj/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'median %.4f' % d['median_ms_per_step'], 'K1 %.4f K2 %.4f' % (d['kernel_ms']['rollout'], d['kernel_ms']['social']), 'clock %.3f' % d['sustained_clock_ghz'], 'frac %.3f' % d['roofline']['frac'])"; }
case $what in
tests)
  python -m pytest ${@:-tests} -x -q -m gpu > $OUT/${TAG}_tests.log 2>&1; echo "tests rc=$?"
  grep -n "passed\|failed" $OUT/${TAG}_tests.log; grep -B5 -A40 "^E " $OUT/${TAG}_tests.log | head -80 ;;
ab)
  WLS=$1; shift
  for rep in 1 2 3; do for v in "${@:--}"; do for w in $WLS; do
    ARGS="--workload $w --no-cpu-baseline --no-extra --no-verify --steps ${STEPS:-30} --warmup 3"
    if [ "$v" = "-" ]; then python bench.py $ARGS 2>/dev/null | tail -1 | line "this tree"
    elif [ -d "$v" ]; then (cd $v && python bench.py $ARGS 2>/dev/null | tail -1) | line "$v"
    else SFW_ALLOW_ABLATION=1 SFW_HIP_LIB=$PWD/$v python bench.py $ARGS 2>/dev/null | tail -1 | line "$v"; fi
  done; done; done | tee $OUT/${TAG}_ab.txt ;;
timeline)
  bash tools/step_timeline.sh ${1:-cfg2} > $OUT/${TAG}_step_timeline_${1:-cfg2}.txt 2>&1; cat $OUT/${TAG}_step_timeline_${1:-cfg2}.txt ;;
cycle)
  N=${1:-300}
  { for o in 0 60 240; do echo "== build/cycle_latency $N $o (the same costmap every cycle: not re-sent)"; build/cycle_latency $N $o; done
    for o in 0 60; do echo "== build/cycle_latency $N $o 0 1 (the costmap changes every cycle: one H2D copy per cycle)"; build/cycle_latency $N $o 0 1; done
    echo "== build/cycle_latency $N 0 1 (marker capture)"; build/cycle_latency $N 0 1
    echo "== SFW_CYCLE_FUSED=0 (three launches per cycle) build/cycle_latency $N 0"; SFW_CYCLE_FUSED=0 build/cycle_latency $N 0
    echo "== SFW_CYCLE_FUSED=0 build/cycle_latency $N 60"; SFW_CYCLE_FUSED=0 build/cycle_latency $N 60; } > $OUT/${TAG}_latency.txt 2>&1
  cat $OUT/${TAG}_latency.txt
  bash tools/cycle_timeline.sh > /dev/null 2>&1; cp $OUT/cycle_timeline.txt $OUT/${TAG}_cycle_timeline.txt; cat $OUT/${TAG}_cycle_timeline.txt ;;
hostgap)
  SFW_DEBUG_STAGE=1 python tools/host_gap_probe.py ${1:-cfg2} ${2:-60} 2>&1 | grep -v "^\[sfw\] stage" | tail -3
  SFW_DEBUG_STAGE=1 python tools/host_gap_probe.py ${1:-cfg2} 12 2>&1 | grep "stage:" | tail -2 ;;
evidence)
  bash tools/gpu_profile_round.sh $TAG
  TAG=$TAG bash tools/session.sh cycle
  TAG=$TAG bash tools/session.sh timeline cfg2 > /dev/null; TAG=$TAG bash tools/session.sh timeline target > /dev/null
  python tools/sweep_parity.py 100 3000 > $OUT/${TAG}_sweep_a.txt 2>&1; tail -1 $OUT/${TAG}_sweep_a.txt
  python tools/sweep_parity.py 20000 3000 > $OUT/${TAG}_sweep_b.txt 2>&1; tail -1 $OUT/${TAG}_sweep_b.txt
  python tools/cycle_k2.py 0,1,5,8,12,20,30,50 0,16,60,120,240,720 > $OUT/${TAG}_cycle_k2.txt 2>&1; tail -9 $OUT/${TAG}_cycle_k2.txt ;;
*) echo "usage: tools/session.sh tests|ab|timeline|cycle|hostgap|evidence ..."; exit 2 ;;
esac
