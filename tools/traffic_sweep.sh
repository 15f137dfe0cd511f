root=$(pwd); cd /tmp; export TMPDIR=/tmp
for task in reach push; do
for n in 64 512 2048 4096 8192 16384; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=$root/gpurun_out/r06_traffic/${task}_${n}_$c
    rm -rf $out; mkdir -p $out
    rocprofv3 --pmc $c --output-format csv -d $out -- python $root/bench.py --task $task --envs-per-gpu $n --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $out/log.txt 2>&1
  done
done
done
cd $root; python - <<'PY'
import csv, glob, collections, json
rows = []
for task in ('reach', 'push'):
    for n in (64, 512, 2048, 4096, 8192, 16384):
        rec = {'task': task, 'envs': n}
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            agg = collections.defaultdict(list)
            for f in glob.glob('gpurun_out/r06_traffic/%s_%d_%s/*/*_counter_collection.csv' % (task, n, c)):
                for r in csv.DictReader(open(f)):
                    if r['Counter_Name'] == c:
                        agg[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
            rec[c] = {k: [len(v), sum(v) / len(v)] for k, v in agg.items()}
        rows.append(rec)
        print(json.dumps(rec))
json.dump(rows, open('gpurun_out/r06_traffic_by_kernel_and_batch.json', 'w'), indent=1)
PY
rm -rf gpurun_out/r06_traffic
