// debug harness: per-phase timing of the substep on env 0 (PMG_PROFILE build, not shipped)
#define PMG_PROFILE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include "pmg_kernels.h"
template <int NB, int MAXC>
__global__ void __launch_bounds__(64, 2) k_prof(pmg::EnvParams P, const float* act) { pmg::step_env<NB, MAXC, false>(P, act, pmg::scheduled_env(P, (int)blockIdx.x)); }
__global__ void __launch_bounds__(64, 2) k_prof_packed(pmg::EnvParams P, const float* act) { long long t0 = __builtin_readcyclecounter(); pmgp::step_group(P, act, (int)blockIdx.x); long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) P.prof[32 + blockIdx.x] = t1 - t0; }
__global__ void __launch_bounds__(64, 2) k_prof_obj4(pmg::EnvParams P, const float* act) { __shared__ pmgp::ObjLds4 sm; pmgp::step_group_obj<false>(P, act, (int)blockIdx.x, sm); }
int main(int argc, char** argv)
{
    int task = argc > 1 ? atoi(argv[1]) : 0;  // 0 reach (tip low), 1 push, 4 block_stack with four blocks on the table
    pmg::EnvParams P; memset(&P, 0, sizeof(P)); P.chest = -1;
    int N = argc > 3 ? atoi(argv[3]) : 4096; P.n_envs = N; P.task = task; P.nb = task == 0 ? 0 : (task == 4 ? 4 : 1); P.multi = task == 4; P.grasping = task == 4; P.has_obj = task != 0; P.max_steps = 50; P.binary_reward = 1;
    P.adim = task == 4 ? 4 : 3; P.odim = task == 0 ? 3 : (task == 4 ? 72 : 20); P.pdim = task == 0 ? 3 : (task == 4 ? 16 : 7); P.gdim = task == 4 ? 12 : 3; P.packed = P.odim + P.pdim + 9; P.thr = 0.05f;
    float lo[3] = {-0.67f, -0.2f, 0.175f}, hi[3] = {-0.37f, 0.2f, 0.55f}, tc[3] = {-0.52f, 0, 0.08f}, th[3] = {0.25f, 0.35f, 0.08f};
    for (int a = 0; a < 3; a++) { P.ee_lo[a] = lo[a]; P.ee_hi[a] = hi[a]; P.table_c[a] = tc[a]; P.table_h[a] = th[a]; }
    P.table_mu = 0.1f; P.near_r = 0.065f; P.wave_budget = 1536;
    std::vector<float> hot(N * 32, 0.f), goal(N * 16, 0.f), blk(N * 13 * 4, 0.f), act(N * 4, 0.f);
    // joint pose with the tip at z ~ 0.176 (push start pose): from the oracle's reset
    float q0[9] = {0.f, -0.4712f, 0.f, 1.9904f, 0.f, -0.6800f, 0.f, 0.035f, 0.035f};
    float zt = argc > 2 ? atof(argv[2]) : 0.176f;
    for (int i = 0; i < N; i++) { for (int d = 0; d < 9; d++) hot[i * 32 + d] = q0[d]; hot[i*32+18] = -0.52f; hot[i*32+19] = 0; hot[i*32+20] = zt; hot[i*32+28] = 0.035f;
        int nbl = task == 4 ? 4 : 1; for (int b = 0; b < nbl; b++) { float* o = &blk[(i*nbl+b)*13]; o[0] = -0.45f - 0.05f*b; o[1] = 0.1f - 0.06f*b; o[2] = 0.175f; o[6] = 1.f; } }
    hipMalloc(&P.hot, hot.size()*4); hipMalloc(&P.cold, N*16*4); { std::vector<float> cold(N*16, 0.f); for (int i = 0; i < N; i++) { cold[i*16+7] = 3.f; for (int b = 0; b < 5; b++) cold[i*16+8+b] = (float)b; } hipMemcpy(P.cold, cold.data(), cold.size()*4, hipMemcpyHostToDevice); } hipMalloc(&P.goal, goal.size()*4); hipMalloc(&P.blocks, blk.size()*4); hipMalloc(&P.out, (size_t)N*P.packed*4);
    { std::vector<int> sc(3 + 3 * N, 0); sc[1] = N; for (int i = 0; i < N; i++) sc[2 + N + i] = i; hipMalloc(&P.sched, sc.size()*4); hipMemcpy(P.sched, sc.data(), sc.size()*4, hipMemcpyHostToDevice); }
    hipMalloc(&P.prof, (32 + 8192)*8); hipMemset(P.prof, 0, (32 + 8192)*8);
    float* dact; hipMalloc(&dact, act.size()*4); hipMemcpy(dact, act.data(), act.size()*4, hipMemcpyHostToDevice);
    hipMemcpy(P.hot, hot.data(), hot.size()*4, hipMemcpyHostToDevice); hipMemcpy(P.blocks, blk.data(), blk.size()*4, hipMemcpyHostToDevice); hipMemset(P.goal, 0, goal.size()*4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(P.prof, 0, 32*8);
        hipEventRecord(a);
        if (task == 0) hipLaunchKernelGGL((k_prof<0, 8>), dim3(N), dim3(64), 0, 0, P, dact); else if (task == 4) hipLaunchKernelGGL((k_prof<5, 48>), dim3(N), dim3(64), 0, 0, P, dact); else hipLaunchKernelGGL((k_prof<1, 24>), dim3(N), dim3(64), 0, 0, P, dact);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long pr[32]; hipMemcpy(pr, P.prof, sizeof(pr), hipMemcpyDeviceToHost);
        printf("rowspace: build %.0f (jacobian %.0f response %.0f coefficients %.0f) solve %.0f (motor rows %.0f normals %.0f frictions %.0f) cycles/substep, %.2f full iterations/substep\n", (pr[11]+pr[22]+pr[23])/100., pr[22]/100., pr[23]/100., pr[11]/100., pr[12]/100., pr[24]/100., pr[25]/100., pr[26]/100., pr[14]/100.);
        if (pr[31]) printf("ik: %lld iterations, cycles/iteration: fk %.0f error %.0f jjt %.0f solve %.0f\n", pr[31], (double)pr[27]/pr[31], (double)pr[28]/pr[31], (double)pr[29]/pr[31], (double)pr[30]/pr[31]);
        printf("task %d rep %d kernel %.3f ms | cycles/substep: fk %.0f detect %.0f dyn %.0f rows %.0f pgs %.0f | whole-step cycles: ik %lld loop %lld out %lld | nc %.0f con %.0f\n", task, rep, ms, pr[0]/100., pr[1]/100., pr[2]/100., pr[3]/100., pr[4]/100., pr[5], pr[6], pr[7], pr[8]/100., pr[9]/100.);
    }
    { long long ph[32]; hipMemcpy(ph, P.prof, sizeof(ph), hipMemcpyDeviceToHost); printf("phase cycles/substep (last rep): collide-narrow %.0f compact %.0f | R1 %.0f R2 %.0f R3 %.0f R4 %.0f\n", ph[16]/100., ph[17]/100., ph[18]/100., ph[19]/100., ph[20]/100., ph[21]/100.); }
    if (task == 0) { // packed path: all envs on the free list, 4 per wave
        std::vector<int> sc(3 + 3 * N, 0); sc[0] = 0; sc[1] = N; for (int i = 0; i < N; i++) sc[2 + N + i] = i;
        hipMemcpy(P.sched, sc.data(), sc.size()*4, hipMemcpyHostToDevice);
        hipMemcpy(P.hot, hot.data(), hot.size()*4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; rep++) { hipMemset(P.prof, 0, 32*8); hipEventRecord(a); hipLaunchKernelGGL(k_prof_packed, dim3((N+3)/4), dim3(64), 0, 0, P, dact); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            long long pr[32]; hipMemcpy(pr, P.prof, sizeof(pr), hipMemcpyDeviceToHost);
            { std::vector<long long> w((N+3)/4); hipMemcpy(w.data(), P.prof + 32, w.size()*8, hipMemcpyDeviceToHost); long long mn = w[0], mx = w[0]; double sm = 0; for (auto v : w) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; sm += v; } printf("per-wave cycles: min %lld mean %.0f max %lld (block 0: %lld)\n", mn, sm / w.size(), mx, w[0]); }
            printf("PACKED kernel %.3f ms | cycles/substep: fk %.0f low+inertia %.0f bias %.0f minv %.0f qdd+rows %.0f pgs-tail %.0f pgs-iters %.0f | per wave-step: ik %.0f loop %.0f\n", ms, pr[0]/100., pr[1]/100., pr[2]/100., pr[3]/100., pr[4]/100., pr[5]/100., pr[6]/100., (double)pr[7], (double)pr[8]); }
    }
    if (task == 1) { // packed one-object path: four envs per wavefront
        hipMemcpy(P.hot, hot.data(), hot.size()*4, hipMemcpyHostToDevice); hipMemcpy(P.blocks, blk.data(), blk.size()*4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; rep++) { hipMemset(P.prof, 0, 32*8); hipEventRecord(a); hipLaunchKernelGGL(k_prof_obj4, dim3((N+3)/4), dim3(64), 0, 0, P, dact); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
            long long pr[32]; hipMemcpy(pr, P.prof, sizeof(pr), hipMemcpyDeviceToHost);
            printf("OBJ4 kernel %.3f ms | cycles/substep: fk %.0f detect %.0f dyn %.0f rows %.0f pgs %.0f | ik %lld loop %lld out %lld | nc %.0f con %.0f\n", ms, pr[0]/100., pr[1]/100., pr[2]/100., pr[3]/100., pr[4]/100., pr[5], pr[6], pr[7], pr[8]/100., pr[9]/100.); }
    }
    std::vector<float> h2(N*32); hipMemcpy(h2.data(), P.hot, h2.size()*4, hipMemcpyDeviceToHost); printf("q0 after: %f %f %f ee z %f\n", h2[1], h2[3], h2[5], h2[20]);
    return 0;
}
