#!/usr/bin/env python3
"""Static instruction census of one kernel by SOURCE LINE (dev tool, no GPU needed).
    hipcc --offload-arch=gfx950 --cuda-device-only -S -gline-tables-only <Makefile flags> pmg_kernels.hip -o k.s
    tools/isa_lines.py k.s <kernel-name-substring> [top]
Every instruction is attributed to the innermost .loc in front of it (inlined code keeps its own file:line); prints the
instruction count per file:function-ish line range, largest first, and totals per file."""
import collections
import re
import sys

path, kern = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = {}
lines = open(path).read().split('\n')
start = None
for i, l in enumerate(lines):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
    if start is None and re.match(r'^_Z\w*%s\w*:' % kern, l):
        start = i
end = next(i for i in range(start, len(lines)) if '.amdhsa_kernel' in lines[i] or lines[i].startswith('.Lfunc_end'))
cur = ('?', 0)
per = collections.Counter()
kinds = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
        continue
    per[cur] += 1
    op = s.split()[0]
    kinds[cur]['dpp' if 'dpp' in s else ('lds' if op.startswith('ds_') else ('salu' if op.startswith('s_') else ('mem' if op.startswith(('global', 'flat', 'scratch', 'buffer')) else 'valu')))] += 1
tot = sum(per.values())
print('kernel %s: %d static instructions' % (kern, tot))
byfile = collections.Counter()
for (f, ln), n in per.items():
    byfile[f] += n
print('per file:', dict(byfile))
for (f, ln), n in per.most_common(top):
    print('%6d  %s:%d  %s' % (n, f, ln, dict(kinds[(f, ln)])))
