#!/bin/bash
# one gpurun call: hardware probe, GPU test tier, kernel A/B over gpurun_ab/*.so, phase timings
mkdir -p gpurun_out/batch
( ./gpurun_ab/probe_dpp.bin ) > gpurun_out/batch/probe.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/batch/gpu_tests.log 2>&1
tail -3 gpurun_out/batch/gpu_tests.log
bash tools/ab.sh "${1:-reach push pick_and_place block_stack chest_push}" > gpurun_out/batch/ab.log 2>&1
cat gpurun_out/batch/probe.log gpurun_out/batch/ab.log
( ./gpurun_ab/prof_k.bin 0 0.176; ./gpurun_ab/prof_k.bin 0 0.30; ./gpurun_ab/prof_k.bin 1 ) > gpurun_out/batch/prof_k.log 2>&1
cat gpurun_out/batch/prof_k.log
