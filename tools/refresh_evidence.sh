#!/bin/bash
# Run on the GPU box (via gpurun) after a kernel change: rocprofv3 kernel stats + counter passes for every task
# (tools/profile.sh), then every committed bench line against those fresh counters.  Everything lands under
# gpurun_out/evidence/ with the names used in profiles/; copy it over with `cp gpurun_out/evidence/* profiles/`.
#   tools/refresh_evidence.sh [round-tag]
set -u
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out/evidence
rm -rf $out; mkdir -p $out
tasks="reach push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place"
for t in $tasks; do
  bash tools/profile.sh $t $tag > /dev/null 2>&1
  cp gpurun_out/profiles/${tag}_${t}4096_* $out/ 2>/dev/null
  cp gpurun_out/profiles/${tag}_${t}4096_* profiles/ 2>/dev/null     # bench.py reads the committed counter passes
done
python bench.py > $out/${tag}_bench_reach4096.json 2>/dev/null
for t in push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place; do
  python bench.py --task $t --steps 100 --warmup 10 > $out/${tag}_bench_${t}4096.json 2>/dev/null
done
python bench.py --task pick_and_place --envs-per-gpu 8192 --dense-reward --steps 100 --warmup 10 > $out/${tag}_bench_pick_and_place8192_dense.json 2>/dev/null
python bench.py --episode-steps 10 --no-cpu-baseline > $out/${tag}_bench_reach4096_short_episodes.json 2>/dev/null
PMG_PACKED=0 python bench.py --no-cpu-baseline > $out/${tag}_bench_reach4096_one_env_per_wave.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_default -- python $root/bench.py > /dev/null 2>&1 )
cp $root/gpurun_out/prof_default/*/*_kernel_stats.csv $out/${tag}_reach4096_default_command_kernel_stats.csv 2>/dev/null
python tools/bench_reward.py > $out/${tag}_reward_kernel.log 2>&1
ls $out | wc -l
