# GPU box: plan radius sweep for the multi-block tasks (PMG_NEAR_R): share of the batch on the full-store list, redo counts, rate
task=${1:-block_stack}
for r in 0.065 0.055 0.05 0.045 0.04; do
  echo "== $task near_r $r"
  PMG_NEAR_R=$r python tools/redo_fraction.py $task 2>&1 | tail -1
  PMG_NEAR_R=$r bash tools/bench_all.sh "" $task | head -1
done
