#!/usr/bin/env python3
"""Where the counter traffic of a batched step goes (review item: "break the 2.2-2.6 x down").

Input: gpurun_out/r06_traffic_by_kernel_and_batch.json -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of
`bench.py --task T --envs-per-gpu N --steps 30 --warmup 5` for N = 64 ... 16 384, averaged PER KERNEL and per launch (the sweep
script is in profiles/r06_traffic_breakdown.txt's header).  Every kernel's traffic is split into a part that does not depend on
the batch (intercept of the 2048 -> 4096-env line: what a launch costs whatever it computes -- its code through the eight
L2s, kernel arguments, the constant tables) and a per-env slope, which is compared with the buffers the kernel touches.  Kernel
code sizes come from the shipped libpmg_hip.so (symbol sizes of the gfx950 code object).

    python tools/traffic_breakdown.py [json] > profiles/r06_traffic_breakdown.txt
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
FETCH_CAL, WRITE_CAL = 1.993, 0.994          # profiles/r05_counter_calibration.json (dword-access kernels; KiB counters)


def kernel_code_sizes(so):
    sizes = {}
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', so, fat])
        data = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(magic, data)]
        for k, p in enumerate(starts):
            part, co = os.path.join(d, 'b%d.bin' % k), os.path.join(d, 'k%d.co' % k)
            open(part, 'wb').write(data[p:starts[k + 1] if k + 1 < len(starts) else len(data)])
            subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + part,
                                   '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
            out = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '-sW', '--demangle', co], capture_output=True, text=True).stdout
            for line in out.splitlines():
                f = line.split(None, 7)
                if len(f) == 8 and f[3] == 'FUNC':
                    name = re.sub(r'\(.*$', '', f[7]).replace('void ', '')
                    sizes[name] = max(sizes.get(name, 0), int(f[2]))
    return sizes


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r06_traffic_by_kernel_and_batch.json')
    rows = json.load(open(path))
    code = kernel_code_sizes(os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc', 'libpmg_hip.so'))
    alg = {'reach': 298, 'push': 486}
    print('# tools/traffic_breakdown.py: FETCH_SIZE / WRITE_SIZE per kernel and per launch against the batch size (KiB as the counters report them; calibrated bytes =')
    print('# FETCH x %.3f, WRITE x %.3f: profiles/r05_counter_calibration.json).  Sweep: gpurun_ab/exp8.sh of the round (rocprofv3 --pmc <counter> --output-format csv' % (FETCH_CAL, WRITE_CAL))
    print('#   -- python bench.py --task T --envs-per-gpu N --steps 30 --warmup 5 --no-cpu-baseline --no-extras; separate passes per counter; 85 launches per kernel).')
    for task in ('reach', 'push'):
        rs = [r for r in rows if r['task'] == task]
        by = {r['envs']: r for r in rs}
        ks = sorted({k for r in rs for k in list(r['FETCH_SIZE']) + list(r['WRITE_SIZE']) if 'rocclr' not in k})
        print('\n== %s: raw KiB per launch, FETCH / WRITE' % task)
        print('%-32s' % 'kernel' + ''.join('%19d' % r['envs'] for r in rs) + '   code B')
        for k in ks:
            print('%-32s' % k[:32] + ''.join('%9.1f /%8.1f' % (r['FETCH_SIZE'].get(k, [0, 0])[1], r['WRITE_SIZE'].get(k, [0, 0])[1]) for r in rs) +
                  '   %6d' % code.get(k, 0))
        # fixed part and slope from the 2048 -> 4096 line (same kernels at both sizes, except the plan, whose form changes at 4096: 4096 -> 8192)
        print('\n   per batched step of 4096 envs, calibrated bytes: fixed part (intercept) + per-env slope x 4096')
        tot_fix = tot_var = 0.0
        step_fix = step_var = 0.0
        for k in ks:
            a, b = (2048, 4096) if all(k in by[n]['FETCH_SIZE'] and by[n]['FETCH_SIZE'][k][0] for n in (2048, 4096)) else (4096, 8192)
            if not all(k in by[n]['FETCH_SIZE'] for n in (a, b)):
                continue
            f = [by[n]['FETCH_SIZE'].get(k, [0, 0])[1] * 1024 * FETCH_CAL for n in (a, b)]
            w = [by[n]['WRITE_SIZE'].get(k, [0, 0])[1] * 1024 * WRITE_CAL for n in (a, b)]
            if by[4096]['FETCH_SIZE'].get(k, [0, 0])[0] == 0:
                continue
            fs, ws = (f[1] - f[0]) / (b - a), (w[1] - w[0]) / (b - a)
            f4 = by[4096]['FETCH_SIZE'][k][1] * 1024 * FETCH_CAL
            w4 = by[4096]['WRITE_SIZE'].get(k, [0, 0])[1] * 1024 * WRITE_CAL
            ffix, wfix = f4 - fs * 4096, w4 - ws * 4096
            print('   %-30s fetch %8.0f fixed + %6.1f B/env   write %8.0f fixed + %6.1f B/env   (code %6d B x 8 L2s = %7d)' %
                  (k[:30], ffix, fs, wfix, ws, code.get(k, 0), 8 * code.get(k, 0)))
            tot_fix += ffix + wfix; tot_var += (fs + ws) * 4096
            if 'step' in k or 'redo' in k:
                step_fix += ffix + wfix; step_var += (fs + ws) * 4096
        A = alg[task] * 4096
        print('   step kernels (+ redo): %.2f MB = %.2f fixed + %.2f per-env  |  algorithmic %.2f MB  ->  %.2f x in all, %.2f x for the per-env part' %
              ((step_fix + step_var) / 1e6, step_fix / 1e6, step_var / 1e6, A / 1e6, (step_fix + step_var) / A, step_var / A))
        print('   every kernel of the step (plan, step, redo, masked reset): %.2f MB = %.2f fixed + %.2f per-env' % ((tot_fix + tot_var) / 1e6, tot_fix / 1e6, tot_var / 1e6))


if __name__ == '__main__':
    main()
