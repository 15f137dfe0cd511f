#!/bin/bash
# instruction-cache counters of a bench configuration (run on the GPU box): tools/icache.sh <task>
task=${1:-push}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/icache_$task; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $out/a -- python $root/bench.py --task $task --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_IFETCH --output-format csv -d $out/b -- python $root/bench.py --task $task --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/b.log 2>&1
cd $root
python - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ('a', 'b'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in glob.glob(out + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:40]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in acc.items():
        if 'k_step' in k: print(sub, k, {c: '%.3g' % x for c, x in v.items()})
PY
tail -3 $out/a.log | cut -c1-300
