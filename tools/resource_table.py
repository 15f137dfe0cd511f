#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel of libpmg_hip.so from hipcc's -Rpass-analysis=kernel-resource-usage
(`make -C pybullet_multigoal_gym_amd/csrc resources`, compile only, no GPU needed):
    tools/resource_table.py [extra hipcc flags, e.g. -DPMG_WAVES_PER_EU=1] > profiles/rNN_kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc')
mk = open(os.path.join(SRC, 'Makefile')).read()
flags = re.search(r'^CXXFLAGS \?= (.*)$', mk, re.M).group(1).split()
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950'] + flags + sys.argv[1:] + ['-c', 'pmg_kernels.hip', '-o', '/dev/null',
                                                                                 '-Rpass-analysis=kernel-resource-usage']
err = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r'remark:\s+([A-Za-z \[\]/]+):\s+(\S+)', line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == 'Function Name':
        cur = {'name': v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
print('# hipcc flags: ' + ' '.join(flags + sys.argv[1:]))
print('%-52s %5s %5s %9s %9s %8s %4s %7s' % ('kernel', 'VGPR', 'AGPR', 'SGPRspill', 'VGPRspill', 'scratchB', 'occ', 'LDS B'))
for r, n in zip(rows, names):
    n = re.sub(r'\(.*$', '', n).replace('void ', '').replace('(anonymous namespace)::', '')
    print('%-52s %5s %5s %9s %9s %8s %4s %7s' % (n[:52], r.get('VGPRs'), r.get('AGPRs'), r.get('SGPRs Spill'), r.get('VGPRs Spill'),
                                                 r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
