#!/bin/bash
# exploration helper: bench value / ms per step for a list of "task:episode_steps" settings
for spec in "$@"; do
  t=${spec%%:*}; ep=${spec##*:}
  python bench.py --task $t --steps 100 --warmup 10 --no-cpu-baseline --episode-steps $ep 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$spec', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done
