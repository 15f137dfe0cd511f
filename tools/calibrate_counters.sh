#!/bin/bash
# Run on the GPU box: calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on kernels with KNOWN byte counts (MI355X_MICROARCH.md
# "HBM": FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950; other widths and WRITE_SIZE are uncalibrated).
#   G = 3  -> pmg_k_reward3      : dwordx4 loads / stores  (16 B per lane)
#   G = 7 / 12 -> pmg_k_reward_flat: dwordx4 loads, dword + byte stores
#   G = 7 with PMG_REWARD_GENERIC=1 -> pmg_k_reward: dword loads, dword + byte stores (the access width of the step kernels)
# Also records the reward kernels under --kernel-trace (their duration by the profiler, not by host timers).
#   tools/calibrate_counters.sh [round-tag]
set -u
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/calib; rm -rf $out; mkdir -p $out $root/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
export PMG_REWARD_LAUNCHES=4
B=$((1<<26))
for G in 3 7 12; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$G -- python $root/tools/bench_reward.py $B $G > $out/trace_$G.json 2> $out/trace_$G.err
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_$G -- python $root/tools/bench_reward.py $B $G > $out/pmc_${c}_$G.json 2> $out/pmc_${c}_$G.err
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  PMG_REWARD_GENERIC=1 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${c}_7d -- python $root/tools/bench_reward.py $B 7 > $out/pmc_${c}_7d.json 2> $out/pmc_${c}_7d.err
done
cd $root && python - "$tag" "$B" <<'PY'
import csv, glob, json, os, sys
tag, B = sys.argv[1], int(sys.argv[2])
root = os.getcwd()
out = os.path.join(root, 'gpurun_out', 'calib')
res = {'items': B, 'kernels': {}}
for G in (3, 7, 12, '7d'):
    per = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        vals = {}
        for f in glob.glob(os.path.join(out, 'pmc_%s_%s' % (c, G), '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'pmg_k_reward' in r.get('Kernel_Name', '') and r['Counter_Name'] == c:
                    vals.setdefault(r['Dispatch_Id'], 0.0)
                    vals[r['Dispatch_Id']] += float(r['Counter_Value'])
        per[c] = sum(vals.values()) / max(1, len(vals))
    dur = None
    label = G
    G = 7 if G == '7d' else G
    for f in glob.glob(os.path.join(out, 'trace_%s' % label, '**', '*kernel_stats.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'pmg_k_reward' in r.get('Name', ''):
                dur = {'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']), 'min_ns': float(r['MinNs']), 'max_ns': float(r['MaxNs'])}
        rows = list(csv.reader(open(f)))
        csv.writer(open(os.path.join(root, 'gpurun_out', 'profiles', '%s_reward_G%s_kernel_stats.csv' % (tag, label)), 'w')).writerows(rows)
    rd, wr = B * 8 * G, B * 5
    k = {'G': G, 'read_bytes': rd, 'write_bytes': wr, 'FETCH_SIZE_KiB': per['FETCH_SIZE'], 'WRITE_SIZE_KiB': per['WRITE_SIZE'],
         'fetch_factor': rd / (per['FETCH_SIZE'] * 1024.0) if per['FETCH_SIZE'] else None,
         'write_factor': wr / (per['WRITE_SIZE'] * 1024.0) if per['WRITE_SIZE'] else None, 'kernel_trace': dur}
    if dur:
        k['GBps_by_kernel_trace'] = (rd + wr) / dur['avg_ns']
        k['frac_of_8TBps'] = k['GBps_by_kernel_trace'] / 8000.0
    res['kernels']['G%s' % label] = k
res['fetch_factor_dwordx4'] = res['kernels']['G3']['fetch_factor']
res['fetch_factor_dword'] = res['kernels']['G7d']['fetch_factor']
res['write_factor_dword'] = res['kernels']['G7d']['write_factor']
res['note'] = 'factor = known bytes / counter bytes on a 64 Mi-item launch far beyond the 256 MiB Infinity Cache; G=3 / 7 / 12 stream dwordx4, G7d (PMG_REWARD_GENERIC=1: pmg_k_reward) dwords'
json.dump(res, open(os.path.join(root, 'gpurun_out', 'profiles', '%s_counter_calibration.json' % tag), 'w'), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != 'kernels'}))
for g, k in res['kernels'].items():
    print(g, {x: k[x] for x in ('fetch_factor', 'write_factor', 'GBps_by_kernel_trace') if x in k})
PY
