#!/bin/bash
# kernel A/B: bench each library build in gpurun_ab/ (plus the in-tree one) on the given tasks
tasks=${1:-push}
mkdir -p gpurun_ab
for t in $tasks; do
  for lib in "" $(ls gpurun_ab/*.so 2>/dev/null); do
    args=""; [ -n "$lib" ] && args="--lib $lib"
    python bench.py --task $t --steps 60 --warmup 10 --no-cpu-baseline --no-extras $args 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', '${lib:-intree}', round(d['value']), round(d['roofline']['kernel_ms'],3))"
  done
done
