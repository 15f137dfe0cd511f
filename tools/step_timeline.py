import csv, glob, sys
f = glob.glob('/tmp/pp/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'pmg_k' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# take steps in the middle
t0 = None
out = []
for r in rows[200:232]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 is None: t0 = s
    print('%-28s start %8.1f us  dur %8.1f us  stream %s' % (r['Kernel_Name'].split('(')[0][-28:], (s - t0) / 1e3, (e - s) / 1e3, r.get('Stream_Id', r.get('Queue_Id'))))
