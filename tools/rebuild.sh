#!/bin/bash
# rebuild the HIP library, the CPU emulator and the phase profiler (dev helper; absolute paths, callable from anywhere)
R=/root/repo
make -C $R/pybullet_multigoal_gym_amd/csrc 2>&1 | grep -E "error" -A5
make -s -C $R/tests/emu 2>&1 | grep -E "error" -A3
mkdir -p $R/gpurun_ab
( cd $R/tools && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../pybullet_multigoal_gym_amd/csrc -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp prof_k.hip -o ../gpurun_ab/prof_k.bin 2>&1 | grep -E "error" -A3 )
ls -la $R/pybullet_multigoal_gym_amd/csrc/libpmg_hip.so | awk '{print $6,$7,$8,$9}'
