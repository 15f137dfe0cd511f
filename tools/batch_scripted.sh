#!/bin/bash
# GPU box: scripted-policy evidence -- teacher-forced single-step parity along the task-solving trajectories (device and
# float32 oracle vs the float64 oracle) and free-running solvability on the device and the oracle.
mkdir -p gpurun_out/scripted
N=${1:-256}
python tools/scripted_suite.py both $N > gpurun_out/scripted/suite.jsonl 2> gpurun_out/scripted/suite.err
for t in pick_and_place push slide block_stack block_rearrange chest_push chest_pick_and_place; do
  T=60; [ $t = push ] && T=300; [ $t = block_stack ] && T=340; [ $t = block_rearrange ] && T=400; [ $t = chest_push ] && T=360; [ $t = chest_pick_and_place ] && T=100
  python tools/teacher_forced.py $t $N $T scripted > gpurun_out/scripted/tf_dev_$t.json 2> gpurun_out/scripted/tf_dev_$t.err
  python tools/teacher_forced.py $t $N $T f32 scripted > gpurun_out/scripted/tf_f32_$t.json 2> gpurun_out/scripted/tf_f32_$t.err
done
python - <<'PY'
import json, glob
print(open('gpurun_out/scripted/suite.jsonl').read())
for f in sorted(glob.glob('gpurun_out/scripted/tf_*.json')):
    try:
        d = json.load(open(f))
    except Exception as ex:
        print(f, 'FAILED', ex); continue
    print(d['task'], d['who'], d['policy'], 'T', d['T'], 'flags', d['flag_mismatches'], '/', d['flags_off_threshold'], 'final success', d['final_success'], 'lists', d['schedule_env_steps'])
    for k, v in d['stats'].items():
        print('   %-12s max %.2e p99.9 %.2e p99 %.2e p50 %.2e  >1e-4: %d  >1e-3: %d' % (k, v['max'], v['p99.9'], v['p99'], v['p50'], v['n_gt_1e-4'], v['n_gt_1e-3']))
PY
