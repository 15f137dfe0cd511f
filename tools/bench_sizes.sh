#!/bin/bash
# GPU box: staggered env-steps/s of every task at several batch sizes (one line per task and size).   tools/bench_sizes.sh [tasks...]
tasks=${*:-reach push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place}
for t in $tasks; do
  for n in 1024 2048 4096 8192 16384 32768; do
    steps=100; [ $n -ge 16384 ] && steps=50
    python bench.py --task $t --envs-per-gpu $n --steps $steps --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-22s x %6d %8.3f M env-steps/s  %8.3f ms/step  kernel min/avg/max %.3f / %.3f / %.3f' % ('$t', $n, d['value'] / 1e6, d['ms_per_step'], r['kernel_ms_min'], r['kernel_ms'], r['kernel_ms_max']))"
  done
done
