/*
 * pmg_kernels.hip -- gfx950 kernels of the batched env: one workgroup = one
 * wavefront = one environment (grid = num_envs, block = 64).
 */
#include <hip/hip_runtime.h>

#include "pmg_kernels.h"
#include "pmg_launch.h"

#ifndef PMG_WAVES_PER_EU
#define PMG_WAVES_PER_EU 2 /* the path is VALU-issue bound from 2 waves/SIMD on (profiles/r01): prefer 256 VGPRs and no spills */
#endif

/* three LDS footprints: reach (no blocks), one object (push / pick_and_place), block_stack (<= 5 blocks) */
template <int NB, int MAXC>
__global__ void __launch_bounds__(64, PMG_WAVES_PER_EU) pmg_k_step(pmg::EnvParams P, const float* __restrict__ actions)
{
    pmg::step_env<NB, MAXC>(P, actions);
}

__global__ void __launch_bounds__(1024) pmg_k_plan(pmg::EnvParams P, const float* __restrict__ actions)
{
    pmg::plan_all(P, actions);
}

__global__ void __launch_bounds__(64) pmg_k_reset(pmg::EnvParams P, const unsigned char* __restrict__ mask)
{
    pmg::reset_env(P, mask);
}

/* _compute_reward on [B, G] batches (HER relabelling): kuka_single_step_base_env.py:237-244.
 * HBM-bound: 2*G*4 bytes in, 5 bytes out per item; one thread per item, rows are contiguous. */
__global__ void __launch_bounds__(256) pmg_k_reward(const float* __restrict__ ag, const float* __restrict__ dg, long long B,
                                                   int G, float thr, int binary, float* __restrict__ reward,
                                                   unsigned char* __restrict__ ok)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float s = 0.f;
    for (int g = 0; g < G; g++) {
        float e = ag[i * G + g] - dg[i * G + g];
        s += e * e;
    }
    float d = sqrtf(s);
    bool na = d > thr;
    if (reward) reward[i] = binary ? (na ? -1.f : -0.f) : -d;
    if (ok) ok[i] = na ? 0 : 1;
}

hipError_t pmg_launch_plan(const pmg::EnvParams& P, const float* d_actions, hipStream_t s)
{
    /* contact tasks: every env is contact-prone, the identity schedule written at create time stays valid */
    if (P.nb == 0 && !P.joint_control) hipLaunchKernelGGL(pmg_k_plan, dim3(1), dim3(pmg::PLAN_THREADS), 0, s, P, d_actions);
    return hipGetLastError();
}
hipError_t pmg_launch_step(const pmg::EnvParams& P, const float* d_actions, hipStream_t s)
{
    if (P.nb == 0) hipLaunchKernelGGL((pmg_k_step<0, 8>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
    else if (P.nb == 1) hipLaunchKernelGGL((pmg_k_step<1, 24>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
    else hipLaunchKernelGGL((pmg_k_step<5, 48>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
    return hipGetLastError();
}
hipError_t pmg_launch_reset(const pmg::EnvParams& P, const unsigned char* d_mask, hipStream_t s)
{
    hipLaunchKernelGGL(pmg_k_reset, dim3(P.n_envs), dim3(64), 0, s, P, d_mask);
    return hipGetLastError();
}
hipError_t pmg_launch_reward(const float* ag, const float* dg, long long B, int G, float thr, int binary, float* reward,
                             unsigned char* ok, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    unsigned grid = (unsigned)((B + 255) / 256);
    hipLaunchKernelGGL(pmg_k_reward, dim3(grid), dim3(256), 0, s, ag, dg, B, G, thr, binary, reward, ok);
    return hipGetLastError();
}
