#!/bin/bash
# LDS-pipe counters of a bench configuration (run on the GPU box): tools/ldspipe.sh <task>
task=${1:-push}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/lds_$task; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $out/a -- python $root/bench.py --task $task --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $out/b -- python $root/bench.py --task $task --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/b.log 2>&1
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
