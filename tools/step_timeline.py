"""Kernel start / duration per step from a rocprofv3 --kernel-trace run (CSV under the directory given, default /tmp/pp):
   rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python bench.py --task push --steps 50 --warmup 5 --no-cpu-baseline --no-extras
   python tools/step_timeline.py [/tmp/pp] [first_row] [rows]"""
import csv, glob, sys
root = sys.argv[1] if len(sys.argv) > 1 else '/tmp/pp'
f = glob.glob(root + '/*/*_kernel_trace.csv')[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 200
count = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'pmg_k' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take steps in the middle
t0 = None
out = []
for r in rows[first:first + count]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 is None: t0 = s
    print('%-28s start %8.1f us  dur %8.1f us  stream %s' % (r['Kernel_Name'].split('(')[0][-28:], (s - t0) / 1e3, (e - s) / 1e3, r.get('Stream_Id', r.get('Queue_Id'))))
