#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh into the small summaries committed under profiles/."""
import csv
import glob
import json
import os
import sys

task, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'r03')
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', 'prof_' + task)
dst = os.path.join(root, 'gpurun_out', 'profiles')
os.makedirs(dst, exist_ok=True)
ALGO = {'reach': 298, 'push': 486, 'slide': 486, 'pick_and_place': 490, 'block_stack': 1246, 'block_rearrange': 1242,
        'chest_push': 1334, 'chest_pick_and_place': 1338}

stats = glob.glob(os.path.join(src, 'trace', '**', '*kernel_stats.csv'), recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(os.path.join(dst, '%s_%s%d_kernel_stats.csv' % (tag, task, N)), 'w') as f:
        csv.writer(f).writerows(rows)
    print('kernel stats:', rows[1][:4] if len(rows) > 1 else rows)

summary = {}
for d in sorted(glob.glob(os.path.join(src, 'pmc_*'))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc, per_kernel = {}, {}
        for r in csv.DictReader(open(f)):
            if 'pmg_k_step' not in r.get('Kernel_Name', '') and 'pmg_k_redo' not in r.get('Kernel_Name', ''):
                continue
            acc.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
            acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
            per_kernel.setdefault(r['Counter_Name'], {}).setdefault(r['Kernel_Name'], set()).add(r['Dispatch_Id'])
        for name, per in acc.items():
            # one batched step may be several dispatches (two contact-store lists, the redo pass): per STEP = the sum
            # over all of them divided by the dispatch count of the most frequent kernel
            steps = max(len(v) for v in per_kernel[name].values())
            summary[name] = {'launches': steps, 'mean_per_launch': sum(per.values()) / steps, 'pass': os.path.basename(d),
                             'kernels': sorted(k.split('(')[0] for k in per_kernel[name])}
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_hash: ties the pass to the kernel sources that were profiled)
SRC = bench.kernel_source_hash()
LIB = bench.library_hash()
summary_out = dict(summary, _stamp={'kernel_source_sha16': SRC, 'library_sha16': LIB, 'tag': tag})
json.dump(summary_out, open(os.path.join(dst, '%s_%s%d_pmc_summary.json' % (tag, task, N)), 'w'), indent=1, sort_keys=True)
if 'FETCH_SIZE' in summary and 'WRITE_SIZE' in summary:
    fk, wk = summary['FETCH_SIZE']['mean_per_launch'], summary['WRITE_SIZE']['mean_per_launch']
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950, other widths and WRITE_SIZE are
    # uncalibrated -- "calibrate on a known byte count in your own access pattern".  tools/calibrate_counters.sh does that
    # with the reward kernels (known bytes; dwordx4 and dword access) and leaves the factors in profiles/<tag>_counter_calibration.json
    cal = {'fetch_dword': 1.0, 'write_dword': 1.0, 'source': 'uncalibrated (no profiles/%s_counter_calibration.json)' % tag}
    for cand in (os.path.join(root, 'gpurun_out', 'profiles', '%s_counter_calibration.json' % tag), os.path.join(root, 'profiles', '%s_counter_calibration.json' % tag)):
        if os.path.exists(cand):
            c = json.load(open(cand))
            if c.get('fetch_factor_dword') and c.get('write_factor_dword'):
                cal = {'fetch_dword': c['fetch_factor_dword'], 'write_dword': c['write_factor_dword'], 'source': os.path.basename(cand)}
                break
    json.dump({'task': task, 'envs_per_gpu': N, 'kernel_source_sha16': SRC, 'library_sha16': LIB,
               'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python bench.py --task %s --envs-per-gpu %d --steps 50 --warmup 5 --no-cpu-baseline --no-extras' % (task, N),
               'kernel': 'pmg_k_step family (+ redo)', 'FETCH_SIZE_KiB': fk, 'WRITE_SIZE_KiB': wk, 'calibration': cal,
               'hbm_bytes_per_launch': (fk * cal['fetch_dword'] + wk * cal['write_dword']) * 1024.0,
               'hbm_bytes_per_launch_raw': (fk + wk) * 1024.0,
               'algorithmic_bytes_per_launch': ALGO[task] * N,
               'note': 'counters are KiB; this kernel moves dwords (lane = element of a state row), so the factors measured on '
                       'the dword-access reward kernel apply, not the x2 of 16 B/lane streams'},
              open(os.path.join(dst, '%s_%s%d_pmc_traffic.json' % (tag, task, N)), 'w'), indent=1)
print(json.dumps({k: round(v['mean_per_launch']) for k, v in summary.items()}))
