#!/bin/bash
# GPU box: issue priority (s_setprio) of the list-0 (low nibble) / list-1 + packed (high nibble) wavefronts.   tools/prio_exp.sh "levels" [tasks...]
levels=${1:-0 3 0x30 0x31}; shift
tasks=${*:-push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place}
for t in $tasks; do
  for pr in $levels; do
    PMG_LIST0_PRIO=$pr python bench.py --task $t --steps 100 --warmup 10 --no-cpu-baseline --no-extras $PRIO_EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-22s prio %-5s %7.3f M  %7.3f ms/step  kernel min/avg/max %.3f / %.3f / %.3f' % ('$t', '$pr', d['value'] / 1e6, d['ms_per_step'], r['kernel_ms_min'], r['kernel_ms'], r['kernel_ms_max']))"
  done
done
