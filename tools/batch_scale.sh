#!/bin/bash
# GPU box: validation at scale (beyond what the -m gpu tier has time for) + phase timings of the final kernels.
#   tools/batch_scale.sh [round-tag]
tag=${1:-r04}
out=gpurun_out/scale; rm -rf $out; mkdir -p $out
# 1. whole-episode error percentiles, device and float32 oracle vs the float64 oracle, 2048 envs x 50 random steps
for t in push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place; do
  python tools/stat_parity.py $t 2048 50 2>/dev/null | tail -1
done > $out/${tag}_stat_parity_2048x50.jsonl
# 2. soak: every task, 2048 envs x 300 steps with resets: non-finite values, objects leaving the workspace
python tools/soak.py 2048 300 > $out/${tag}_soak_2048x300.jsonl 2>/dev/null
# 3. scripted controllers at 1024 envs, device and oracle
python tools/scripted_suite.py both 1024 > $out/${tag}_scripted_suite_1024.jsonl 2>/dev/null
# 4. per-phase cycles of the one-wavefront kernels (tools/prof_k.hip, PMG_PROFILE build)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pybullet_multigoal_gym_amd/csrc -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp tools/prof_k.hip -o /tmp/prof_k 2> $out/prof_k_build.log
( /tmp/prof_k 0 0.176; /tmp/prof_k 0 0.30; /tmp/prof_k 1 0.176; /tmp/prof_k 4 0.176; /tmp/prof_k 4 0.30 ) > $out/${tag}_prof_k_phase_cycles.txt 2>&1
cat $out/*.jsonl $out/${tag}_prof_k_phase_cycles.txt | cut -c1-400
