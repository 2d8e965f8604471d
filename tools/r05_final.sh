#!/bin/bash
# Round 5, the evidence session on the final code — ONCE: GPU suite, default bench line, rocprofv3 stats + PMC passes (target,
# cfg2 with and without the split launch, cfg2 + 64 points, target + 720 points), control-cycle latency, the two 3000-seed
# parity sweeps, the control-cycle K2 grid against round 3's kernels.
bash tools/gpu_profile_round.sh r05
python tools/sweep_parity.py 100 3000 > gpurun_out/r05_sweep_a.txt 2>&1; tail -1 gpurun_out/r05_sweep_a.txt
python tools/sweep_parity.py 20000 3000 > gpurun_out/r05_sweep_b.txt 2>&1; tail -1 gpurun_out/r05_sweep_b.txt
NS=0,1,5,8,12,20,30,50; OS=0,16,60,120,240,720
{ echo "== round 5 (final)"; python tools/cycle_k2.py $NS $OS
  echo "== build/libsfw_hip_soz1.so (round 3)"; SFW_HIP_LIB=build/libsfw_hip_soz1.so python tools/cycle_k2.py $NS $OS
  echo "== round 5 (final, again)"; python tools/cycle_k2.py $NS $OS; } > gpurun_out/r05_cycle_k2_final.txt 2>&1
tail -9 gpurun_out/r05_cycle_k2_final.txt
