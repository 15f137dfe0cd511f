#!/bin/bash
# GPU box: pick_and_place at 8192 envs (BASELINE configs[3]): promotion budget and list-0 priority
for cfg in "8 1536 -1" "8 3072 -1" "8 3072 0" "8 4096 -1" "0 1536 -1"; do
  set -- $cfg
  PMG_FD_DIV=$1 PMG_WAVE_BUDGET=$2 PMG_LIST0_PRIO=$3 python bench.py --task pick_and_place --envs-per-gpu 8192 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('pnp x8192 fd_div %s budget %s prio %s %7.3f M  %7.3f ms/step  kernel min/avg/max %.3f / %.3f / %.3f' % ('$1', '$2', '$3', d['value'] / 1e6, d['ms_per_step'], r['kernel_ms_min'], r['kernel_ms'], r['kernel_ms_max']))"
done
