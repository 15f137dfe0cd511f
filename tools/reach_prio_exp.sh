#!/bin/bash
# GPU box: issue priority of reach's contact-prone wavefronts at several batch sizes
for n in 4096 8192 16384 32768; do
  for pr in 0 1; do
    PMG_LIST0_PRIO=$pr python bench.py --envs-per-gpu $n --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('reach x %6d prio %s %7.3f M  %7.3f ms/step  kernel min/avg/max %.3f / %.3f / %.3f' % ($n, '$pr', d['value'] / 1e6, d['ms_per_step'], r['kernel_ms_min'], r['kernel_ms'], r['kernel_ms_max']))"
  done
done
