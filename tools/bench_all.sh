#!/bin/bash
# GPU box: one line per task -- staggered (value) and lockstep rates, kernel min / avg / max.   tools/bench_all.sh [lib.so] [tasks...]
lib=${1:-}; shift
tasks=${*:-reach push slide pick_and_place block_stack block_rearrange chest_push chest_pick_and_place}
for t in $tasks; do
  for mode in "" "--lockstep"; do
    python bench.py --task $t --steps 100 --warmup 10 --no-cpu-baseline --no-extras $mode ${lib:+--lib $lib} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-22s %-9s %7.3f M  %7.3f ms/step  kernel min/avg/max %.3f / %.3f / %.3f' % ('$t', 'lockstep' if '$mode' else 'staggered', d['value'] / 1e6, d['ms_per_step'], r['kernel_ms_min'], r['kernel_ms'], r['kernel_ms_max']))"
  done
done
