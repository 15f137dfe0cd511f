#!/usr/bin/env python3
"""Build-time check of the hand-written DPP blocks of pmg_wave.h (fma2_bcast_r0_c, wr::fma2_bcast_c, half_fma_bcast_c).

The compiler's hazard recognizer does not look inside inline asm.  Each block carries its own `s_nop 1` for the
2-wait-state "VALU writes a VGPR, DPP reads it" hazard; the 5-wait-state hazard "VALU writes EXEC (v_cmpx*), then a DPP
op" would need more, and nothing in the source prevents the scheduler from placing such a write right in front of a
block.  This script compiles the kernels to gfx950 ISA text (device only, no GPU needed) and walks back from every
ASMSTART that contains a *_dpp instruction: an EXEC write by a VALU op (v_cmpx*) -- or by any instruction naming exec
as its destination -- within the 5 wait states before the first DPP op fails the check.
    tools/check_dpp_hazards.py            # the shipped libpmg_hip.so, disassembled: every DPP instruction (seconds)
    tools/check_dpp_hazards.py --compile  # the hand-written blocks by their ASM markers in freshly compiled ISA text (~75 s)
    exit code 0 = no such sequence"""
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


# ---------------------------------------------------------------------------------------------------------------------
# The same two hazards on the SHIPPED binary (seconds instead of a recompile): the gfx950 code objects are cut out of
# libpmg_hip.so's .hip_fatbin section, disassembled with llvm-objdump, and EVERY DPP instruction -- the compiler's own and
# the hand-written ones alike -- is checked: no VALU write of EXEC inside its 5 wait states, no VALU write of its DPP source
# register inside its 2 wait states (a linear walk in address order, as above).
LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(so_path):
    import tempfile
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    text = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', so_path, fat])
        data = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(magic, data)]
        for k, p in enumerate(starts):                      # one bundle per translation unit
            part, co = os.path.join(d, 'b%d.bin' % k), os.path.join(d, 'k%d.co' % k)
            open(part, 'wb').write(data[p:starts[k + 1] if k + 1 < len(starts) else len(data)])
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + part,
                                   '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
            text.append(subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', co], capture_output=True,
                                       text=True, check=True).stdout)
    return '\n'.join(text)


def _regs(operand):
    """VGPR numbers an operand names: v7 -> {7}, v[10:11] -> {10, 11}; anything else -> {}"""
    m = re.match(r'v(\d+)$', operand)
    if m:
        return {int(m.group(1))}
    m = re.match(r'v\[(\d+):(\d+)\]$', operand)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def check_binary(text):
    """-> (DPP instructions, EXEC-write hazards, source-register hazards, VALU instructions that write EXEC at all)"""
    code = []
    for l in text.splitlines():
        l = l.split('//')[0].strip()
        if not l or l.endswith(':') or l.startswith(('.', 'Disassembly', '/')) or 'file format' in l:
            continue
        code.append(l)
    ndpp = bad_exec = bad_src = valu_exec = 0
    for k, ins in enumerate(code):
        op = ins.split()[0]
        if op.startswith('v_') and writes_exec(ins):
            valu_exec += 1
        if '_dpp' not in op:
            continue
        ndpp += 1
        ops = [o.strip() for o in ins[len(op):].split(',')]
        src = _regs(ops[1].split()[0]) if len(ops) > 1 else set()   # the operand the DPP control applies to
        ws, back = 0, k - 1
        while ws < 5 and back >= 0:
            prev = code[back]
            pop = prev.split()[0]
            if pop.startswith('v_') and writes_exec(prev):
                bad_exec += 1
                print('HAZARD (EXEC, %d wait states): "%s" ahead of "%s"' % (ws, prev, ins))
                break
            if ws < 2 and pop.startswith('v_') and not pop.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                dst = _regs(prev[len(pop):].split(',')[0].strip())
                if dst & src:
                    bad_src += 1
                    print('HAZARD (DPP source, %d wait states): "%s" ahead of "%s"' % (ws, prev, ins))
                    break
            ws += wait_states(prev)
            back -= 1
    return ndpp, bad_exec, bad_src, valu_exec


if __name__ == '__main__':
    if '--compile' in sys.argv:                              # the source-level check: hand-written blocks by their ASM markers
        blocks, bad = check(isa_text())
        print('%d hand-written DPP blocks in the gfx950 ISA, %d with an EXEC write inside the 5-wait-state window' % (blocks, bad))
        sys.exit(1 if bad or blocks == 0 else 0)
    so = os.path.join(SRC, 'libpmg_hip.so')
    ndpp, bad_exec, bad_src, valu_exec = check_binary(disassemble(so))
    print('%s: %d DPP instructions, %d behind an EXEC write (5 wait states), %d behind a write of their source (2 wait states); '
          '%d VALU instructions write EXEC at all' % (os.path.basename(so), ndpp, bad_exec, bad_src, valu_exec))
    sys.exit(1 if bad_exec or bad_src or ndpp == 0 else 0)
