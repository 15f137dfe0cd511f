#!/bin/bash
# teacher-forced calibration: device and float32 oracle vs the float64 oracle, all contact tasks
mkdir -p gpurun_out/tf
for t in reach push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place; do
  python tools/teacher_forced.py $t 1024 50 > gpurun_out/tf/dev_$t.json 2> gpurun_out/tf/dev_$t.err
  python tools/teacher_forced.py $t 1024 50 f32 > gpurun_out/tf/f32_$t.json 2> gpurun_out/tf/f32_$t.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/tf/*.json')):
    try:
        d = json.load(open(f))
    except Exception as ex:
        print(f, 'FAILED', ex); continue
    print(d['task'], d['who'], 'flags', d['flag_mismatches'], '/', d['flags_off_threshold'])
    for k, v in d['stats'].items():
        print('   %-12s max %.2e p99.9 %.2e p99 %.2e p50 %.2e  >1e-4: %d  >1e-3: %d' % (k, v['max'], v['p99.9'], v['p99'], v['p50'], v['n_gt_1e-4'], v['n_gt_1e-3']))
PY
