for cr in 0.064 0.032 0.02; do for r in 0.05 0.04; do
  echo "== chest_reach $cr near_r $r"
  PMG_CHEST_REACH=$cr PMG_NEAR_R=$r python tools/redo_fraction.py chest_push 2>&1 | tail -1
  PMG_CHEST_REACH=$cr PMG_NEAR_R=$r bash tools/bench_all.sh "" chest_push | head -1
  PMG_CHEST_REACH=$cr PMG_NEAR_R=$r bash tools/bench_all.sh "" chest_pick_and_place | head -1
done; done
