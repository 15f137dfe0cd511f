#!/bin/bash
# Run on the GPU box (via gpurun) after a kernel change: counter calibration, rocprofv3 kernel stats + counter passes for
# every task (tools/profile.sh), then every committed bench line against those fresh counters.  Everything lands under
# gpurun_out/evidence/ with the names used in profiles/; copy it over with `cp gpurun_out/evidence/* profiles/`.
#   tools/refresh_evidence.sh [round-tag] [tasks...]
set -u
tag=${1:-r04}
shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out/evidence
rm -rf $out; mkdir -p $out $root/gpurun_out/profiles
tasks=${*:-reach push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place}
# 1. FETCH_SIZE / WRITE_SIZE calibration on kernels with known byte counts + the reward kernels under --kernel-trace
bash tools/calibrate_counters.sh $tag > $out/${tag}_calibration.log 2>&1
cp gpurun_out/profiles/${tag}_counter_calibration.json gpurun_out/profiles/${tag}_reward_G*_kernel_stats.csv $out/ 2>/dev/null
cp gpurun_out/profiles/${tag}_counter_calibration.json profiles/ 2>/dev/null     # profile_summarise.py applies the factors
# 2. per-task kernel stats and counters
for t in $tasks; do
  bash tools/profile.sh $t $tag > /dev/null 2>&1
  cp gpurun_out/profiles/${tag}_${t}4096_* $out/ 2>/dev/null
  cp gpurun_out/profiles/${tag}_${t}4096_* profiles/ 2>/dev/null     # bench.py reads the committed counter passes
done
# 2b. BASELINE config C4: pick_and_place x 8192 envs (kernel stats + counter passes, so that its bench line carries measured traffic)
bash tools/profile.sh pick_and_place $tag 8192 > /dev/null 2>&1
cp gpurun_out/profiles/${tag}_pick_and_place8192_* $out/ 2>/dev/null
cp gpurun_out/profiles/${tag}_pick_and_place8192_* profiles/ 2>/dev/null
# 3. the bench lines (headline = the default command)
python bench.py > $out/${tag}_bench_reach4096.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_reach4096_driver_command.json 2>/dev/null
for t in $tasks; do
  [ $t = reach ] && continue
  python bench.py --task $t --steps 100 --warmup 10 > $out/${tag}_bench_${t}4096.json 2>/dev/null
done
python bench.py --task pick_and_place --envs-per-gpu 8192 --dense-reward --steps 100 --warmup 10 > $out/${tag}_bench_pick_and_place8192_dense.json 2>/dev/null
python bench.py --task pick_and_place --envs-per-gpu 8192 --steps 100 --warmup 10 --no-cpu-baseline > $out/${tag}_bench_pick_and_place8192_binary.json 2>/dev/null
python bench.py --episode-steps 10 --no-cpu-baseline --no-extras > $out/${tag}_bench_reach4096_short_episodes.json 2>/dev/null
python bench.py --lockstep --no-cpu-baseline > $out/${tag}_bench_reach4096_lockstep.json 2>/dev/null
PMG_REACH_TWO_WAVES=0 python bench.py --no-cpu-baseline --no-extras > $out/${tag}_bench_reach4096_one_wave_workgroups.json 2>/dev/null
bash tools/bench_all.sh > $out/${tag}_bench_all_tasks_staggered_and_lockstep.txt 2>&1
PMG_PACKED=0 python bench.py --no-cpu-baseline --no-extras > $out/${tag}_bench_reach4096_one_env_per_wave.json 2>/dev/null
for n in 8192 16384 32768 65536 131072; do
  python bench.py --envs-per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline --no-extras > $out/${tag}_bench_reach${n}.json 2>/dev/null
done
PMG_BENCH_FORCE_DIST=1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/${tag}_bench_reach4096_one_rank_rccl_path.json 2>/dev/null
# 3b. scripted-policy evidence: solvability on device and oracle, teacher-forced parity along the scripted trajectories
bash tools/batch_scripted.sh 256 > $out/${tag}_scripted_batch.log 2>&1
cp gpurun_out/scripted/suite.jsonl $out/${tag}_scripted_suite.jsonl 2>/dev/null
python - <<'PY' > $out/${tag}_scripted_teacher_forced.json
import json, glob, os
out = {}
for f in sorted(glob.glob('gpurun_out/scripted/tf_*.json')):
    try:
        d = json.load(open(f))
    except Exception:
        continue
    out[os.path.basename(f)[3:-5]] = {k: d[k] for k in ('task', 'who', 'policy', 'N', 'T', 'flag_mismatches', 'flags_off_threshold', 'final_success', 'schedule_env_steps', 'stats')}
print(json.dumps(out, indent=1))
PY
# 4. rocprofv3 --kernel-trace --stats of the DEFAULT command (the one the driver's line comes from)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_default -- python $root/bench.py --no-cpu-baseline > /dev/null 2>&1 )
cp $root/gpurun_out/prof_default/*/*_kernel_stats.csv $out/${tag}_reach4096_default_command_kernel_stats.csv 2>/dev/null
python tools/resource_table.py > $out/${tag}_kernel_resources.txt 2>/dev/null
ls $out | wc -l
