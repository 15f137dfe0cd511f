#!/bin/bash
# A/B builds of the HIP library with extra -D flags into gpurun_ab/<name>.so (they travel to the GPU box; tools/ab.sh benches them)
#   tools/build_variant.sh <name> -DPMG_X=1 [...]
R=/root/repo
name=$1; shift
mkdir -p $R/gpurun_ab
cd $R/pybullet_multigoal_gym_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wall -Wno-unused-function \
  -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp "$@" \
  -shared -o $R/gpurun_ab/$name.so pmg_api.cpp pmg_kernels.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | grep -E "error" -A5
ls -la $R/gpurun_ab/$name.so | awk '{print $5,$9}'
