#!/bin/bash
# GPU box: the plan's promotion rule (fingers-down class -> list 0) at other thresholds / wave budgets.   tools/promote_exp.sh [tasks...]
tasks=${*:-push slide pick_and_place}
for t in $tasks; do
  for cfg in "8 1536" "4 1536" "4 2048" "3 2560" "2 3072" "0 1536"; do
    set -- $cfg
    PMG_FD_DIV=$1 PMG_WAVE_BUDGET=$2 python bench.py --task $t --steps 100 --warmup 10 --no-cpu-baseline --no-extras $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('%-16s fd_div %s budget %s %7.3f M  %7.3f ms/step  kernel min/avg/max %.3f / %.3f / %.3f' % ('$t', '$1', '$2', d['value'] / 1e6, d['ms_per_step'], r['kernel_ms_min'], r['kernel_ms'], r['kernel_ms_max']))"
  done
done
