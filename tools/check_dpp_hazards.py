#!/usr/bin/env python3
"""Build-time check of the hand-written DPP blocks of pmg_wave.h (fma2_bcast_r0_c, wr::fma2_bcast_c, half_fma_bcast_c).

The compiler's hazard recognizer does not look inside inline asm.  Each block carries its own `s_nop 1` for the
2-wait-state "VALU writes a VGPR, DPP reads it" hazard; the 5-wait-state hazard "VALU writes EXEC (v_cmpx*), then a DPP
op" would need more, and nothing in the source prevents the scheduler from placing such a write right in front of a
block.  This script compiles the kernels to gfx950 ISA text (device only, no GPU needed) and walks back from every
ASMSTART that contains a *_dpp instruction: an EXEC write by a VALU op (v_cmpx*) -- or by any instruction naming exec
as its destination -- within the 5 wait states before the first DPP op fails the check.
    tools/check_dpp_hazards.py            # exit code 0 = no such sequence in the shipped kernels"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc')


def isa_text():
    mk = open(os.path.join(SRC, 'Makefile')).read()
    flags = re.search(r'^CXXFLAGS \?= (.*)$', mk, re.M).group(1).split()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '--cuda-device-only', '-S'] + flags +
                              ['pmg_kernels.hip', '-o', out], cwd=SRC, stderr=subprocess.DEVNULL)
        return open(out).read()


def wait_states(ins):
    m = re.match(r's_nop\s+(\d+)', ins)
    return int(m.group(1)) + 1 if m else 1


def writes_exec(ins):
    if ins.startswith('v_cmpx'):
        return True
    m = re.match(r'(\S+)\s+([^,\s]+)', ins)          # first operand = destination
    return bool(m) and m.group(2).startswith('exec') and not m.group(1).startswith(('s_cbranch', 's_waitcnt'))


def check(text):
    lines = [l.strip() for l in text.splitlines()]
    code = []           # (index, instruction) of real instructions, plus ASM markers
    for i, l in enumerate(lines):
        if not l or l.startswith(('.', '//')) or l.endswith(':'):
            continue
        if l.startswith(';'):
            if 'ASMSTART' in l or 'ASMEND' in l:
                code.append((i, l))
            continue
        code.append((i, l.split(';')[0].strip()))
    blocks = bad = 0
    for k, (i, l) in enumerate(code):
        if 'ASMSTART' not in l:
            continue
        body = []
        j = k + 1
        while j < len(code) and 'ASMEND' not in code[j][1]:
            body.append(code[j][1])
            j += 1
        if not any('_dpp' in b for b in body):
            continue
        blocks += 1
        inside = 0          # wait states the block itself puts in front of its first DPP op
        for b in body:
            if '_dpp' in b:
                break
            inside += wait_states(b)
        need, back = 5 - inside, k - 1
        while need > 0 and back >= 0:
            ins = code[back][1]
            if 'ASM' in ins:
                back -= 1
                continue
            if writes_exec(ins):
                bad += 1
                print('HAZARD: line %d: "%s" is %d wait state(s) ahead of the DPP block at line %d' % (code[back][0] + 1, ins, 5 - need + inside, i + 1))
                break
            need -= wait_states(ins)
            back -= 1
    return blocks, bad


if __name__ == '__main__':
    blocks, bad = check(isa_text())
    print('%d hand-written DPP blocks in the gfx950 ISA, %d with an EXEC write inside the 5-wait-state window' % (blocks, bad))
    sys.exit(1 if bad or blocks == 0 else 0)
