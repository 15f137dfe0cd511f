#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 evidence for one bench configuration.
#   tools/profile.sh <task> [round-tag] [envs-per-gpu]
# 1. --kernel-trace --stats of `python bench.py --task T --steps 50 --warmup 5 --no-cpu-baseline --no-extras`
# 2. separate --pmc passes (never combined with traces): FETCH_SIZE, WRITE_SIZE, SQ instruction / wait / fp32-op counters
# Summaries land in gpurun_out/prof_<task>/ ; tools/profile_summarise.py turns them into profiles/<tag>_<task><N>_*.
set -u
task=${1:-reach}; tag=${2:-r04}; n=${3:-4096}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$task
rm -rf $out; mkdir -p $out
cmd="python $root/bench.py --task $task --envs-per-gpu $n --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $cmd > $out/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $out/pmc_$name -- $cmd > $out/pmc_$name.log 2>&1
done
cd $root && python tools/profile_summarise.py $task $tag $n
