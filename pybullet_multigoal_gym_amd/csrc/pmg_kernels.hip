/*
 * pmg_kernels.hip -- gfx950 kernels of the batched env.  A workgroup is always ONE wavefront (block = 64): it
 * simulates one environment (lane = link / DoF) or, on the fast paths, four (one per 16-lane DPP row);
 * pmg_k_plan decides per step which envs go where (DESIGN.md sections 3.1-3.1c).
 */
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "pmg_kernels.h"
#include "pmg_launch.h"

#ifndef PMG_WAVES_PER_EU
#define PMG_WAVES_PER_EU 2 /* 256 VGPRs and no spills in the hot loops; 3-4 waves/SIMD (168 / 128 VGPRs, spills) measured no faster */
#endif

/* three LDS footprints: reach (no blocks), one object (push / pick_and_place / slide), block_stack (<= 5 blocks);
 * CYL: the one object is the slide puck (cylinder x box pairs, anisotropic inertia) */
template <int NB, int MAXC, int CYL>
__global__ void __launch_bounds__(64, PMG_WAVES_PER_EU) pmg_k_step(pmg::EnvParams P, const float* __restrict__ actions)
{
    pmg::step_env<NB, MAXC, CYL>(P, actions, pmg::scheduled_env(P, (int)blockIdx.x));
}

/* reach, tip control: workgroups [0, n_prone) run the contact-prone list one env per wavefront (the slow waves get the
 * lowest ids and start first), the next ceil(n_free / 4) run the contact-free list four envs per wavefront; the
 * grid is sized for the worst case (N) and its surplus workgroups, all at the END of the dispatch order so that they
 * cannot unbalance the placement of the real ones, exit on their first instruction */
/* the device-side choice between the two reach kernels: two wavefronts per workgroup as soon as the step has a
 * contact-prone env (a threshold of 1 / 64 of the batch measured worse: one env on the table already sets the step time) */
/* Registers of the reach kernels.  Two wavefronts per SIMD share the 512-entry unified file: 256 registers each, ArchVGPRs
 * and AccVGPRs together.  Left alone the allocator takes all 256 as ArchVGPRs and spills the rest to SCRATCH memory -- and
 * a scratch reload inside the substep loop is a memory round trip on the one serial chain (a build with 130 spilled
 * registers ran at 2.4 M env-steps/s against 3.8 M).  Capping the ArchVGPRs leaves the remainder of the 256 as AccVGPRs,
 * and the allocator spills there instead (v_accvgpr_write / _read: one VALU instruction, no memory). */
#ifndef PMG_REACH_NUM_VGPR
#define PMG_REACH_NUM_VGPR 232
#endif
#if PMG_REACH_NUM_VGPR > 0 && defined(__HIP_DEVICE_COMPILE__)
#define PMG_REACH_VGPRS __attribute__((amdgpu_num_vgpr(PMG_REACH_NUM_VGPR)))
#else
#define PMG_REACH_VGPRS
#endif
__device__ __forceinline__ bool two_wave_step(int n_prone, int n_envs) { return n_prone > 0; }
__global__ void __launch_bounds__(64, PMG_WAVES_PER_EU) PMG_REACH_VGPRS pmg_k_step_reach(pmg::EnvParams P, const float* __restrict__ actions, int defer)
{
    const int b = (int)blockIdx.x, n0 = P.sched[0];
    if (defer && two_wave_step(n0, P.n_envs)) return;       /* this step belongs to the two-wavefront kernel */
    if (b < n0) pmg::step_env<0, 8, false>(P, actions, P.sched[2 + b]);
    else pmgp::step_group(P, actions, b - n0);
}
/* The same with TWO wavefronts per workgroup, for steps that have contact-prone envs (a batched step lasts as long as
 * its slowest wavefront, and that is one of them): on the contact-prone list wavefront 1 runs the narrowphase of every
 * substep beside wavefront 0's dynamics (helper_wave_loop: 32 k -> 27 k cycles per substep with both fingers on the
 * table); on the contact-free list both wavefronts carry four envs each, so that no wavefront of the launch idles.
 * 4096 envs, staggered episodes: 1.34 -> 1.20 ms per batched step.  A batch WITHOUT contact-prone envs is better off
 * with one-wavefront workgroups (the dispatcher places 1024 of them exactly one per SIMD: 0.54 ms; 512 two-wave ones:
 * 0.82 ms), and beyond one packed wavefront per SIMD the second wave slot of every contact-prone env costs more than the
 * shorter chain saves (16 384 envs: 7.4 vs 8.0 M env-steps/s).  So up to 4096 envs per 256 CUs BOTH kernels are launched
 * every step and the device picks: each reads the plan's contact-prone count on its first instruction and the one whose
 * turn it is not exits (an empty launch: ~5 us of a 1.2 ms step; the host cannot know the count without a sync, and a
 * device-resident rollout queues its steps far ahead of the GPU).  Variants tried:
 * the idle second wavefront of the contact-free workgroups exiting at once (0.82 ms again); the two lists as two
 * launches on two streams (1.44 ms). */
__global__ void __launch_bounds__(128, PMG_WAVES_PER_EU) PMG_REACH_VGPRS pmg_k_step_reach2(pmg::EnvParams P, const float* __restrict__ actions)
{
    const int b = (int)blockIdx.x, n0 = P.sched[0];
    if (!two_wave_step(n0, P.n_envs)) return;               /* a (nearly) contact-free step: the one-wavefront kernel's turn */
    if (b < n0) pmg::step_env<0, 8, false, true>(P, actions, P.sched[2 + b]);
    else pmgp::step_group(P, actions, 2 * (b - n0) + ((int)threadIdx.x >> 6));
}
/* envs the packed path gave up on (a finger reached the table although the plan said it would not) */
__global__ void __launch_bounds__(64, PMG_WAVES_PER_EU) PMG_REACH_VGPRS pmg_k_redo(pmg::EnvParams P, const float* __restrict__ actions)
{
    const int* redo = P.sched + 2 + 2 * P.n_envs;
    if ((int)blockIdx.x >= redo[0]) return;
    pmg::step_env<0, 8, false>(P, actions, redo[1 + blockIdx.x]);
}

__global__ void __launch_bounds__(1024) pmg_k_plan(pmg::EnvParams P, const float* __restrict__ actions)
{
    pmg::plan_all(P, actions);
}
__global__ void __launch_bounds__(1024) pmg_k_plan_reach(pmg::EnvParams P, const float* __restrict__ actions)
{
    pmg::plan_all<true>(P, actions);
}

__global__ void __launch_bounds__(1024) pmg_k_plan_count(pmg::EnvParams P, const float* __restrict__ actions)
{
    pmg::plan_count(P, actions);
}
__global__ void __launch_bounds__(1024) pmg_k_plan_scatter(pmg::EnvParams P, const float* __restrict__ actions)
{
    pmg::plan_scatter(P, actions);
}

__global__ void __launch_bounds__(64) pmg_k_reset(pmg::EnvParams P, const unsigned char* __restrict__ mask, int done_only)
{
    pmg::reset_env(P, mask, done_only);
}

/* _compute_reward on [B, G] batches (HER relabelling): kuka_single_step_base_env.py:237-244.
 * HBM-bound: 2*G*4 bytes in, 5 bytes out per item, each touched once -- so every access is non-temporal (streams past
 * the caches).  G == 3 (every single-object task): a workgroup owns 256 quads of 4 items = 3 x 256 float4 per array,
 * read as three fully coalesced float4 sweeps (lane = consecutive 16 bytes) into LDS; thread t then takes the three
 * float4 of ITS quad from LDS (stride 3: conflict-free), computes four rewards and stores one float4 of rewards and one
 * dword of flags, contiguously.  Measured (tools/reward_variants.hip, 64 Mi pairs): 6.1-6.3 TB/s = the float4-copy
 * ceiling of the part (MI355X_MICROARCH.md: 6.29), against 5.0-5.6 for the thread-owns-three-strided-float4 version
 * of rounds 1-2 (with or without non-temporal stores, one or two quads in flight). */
__global__ void __launch_bounds__(256) pmg_k_reward3(const float4* __restrict__ ag, const float4* __restrict__ dg, long long quads,
                                                    float thr, int binary, float4* __restrict__ reward,
                                                    unsigned int* __restrict__ ok)
{
    __shared__ float4 sa[3 * 256], sd[3 * 256];
    const int t = (int)threadIdx.x;
    for (long long base = (long long)blockIdx.x * 256; base < quads; base += (long long)gridDim.x * 256) {
        const long long n = quads - base < 256 ? quads - base : 256;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const long long w = k * 256 + t;
            if (w < 3 * n) { sa[w] = nt::load4(&ag[3 * base + w]); sd[w] = nt::load4(&dg[3 * base + w]); }
        }
        __syncthreads();
        if (t < n) {
            const float4 a0 = sa[3 * t], a1 = sa[3 * t + 1], a2 = sa[3 * t + 2], d0 = sd[3 * t], d1 = sd[3 * t + 1], d2 = sd[3 * t + 2];
            float e[12] = {a0.x - d0.x, a0.y - d0.y, a0.z - d0.z, a0.w - d0.w, a1.x - d1.x, a1.y - d1.y,
                           a1.z - d1.z, a1.w - d1.w, a2.x - d2.x, a2.y - d2.y, a2.z - d2.z, a2.w - d2.w};
            float r[4];
            unsigned int flags = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float d = sqrtf(e[3 * i] * e[3 * i] + e[3 * i + 1] * e[3 * i + 1] + e[3 * i + 2] * e[3 * i + 2]);
                bool na = d > thr;
                r[i] = binary ? (na ? -1.f : -0.f) : -d;
                flags |= (na ? 0u : 1u) << (8 * i);
            }
            if (reward) nt::store4(make_float4(r[0], r[1], r[2], r[3]), &reward[base + t]);
            if (ok) nt::store(flags, &ok[base + t]);
        }
        __syncthreads();
    }
}
/* multi-block goals (G = 3 * num_block, up to 19 with the gripper tail): a workgroup owns 256 consecutive items = one
 * contiguous span of 256 * G floats = 64 * G float4 per array WHATEVER G is.  Its threads read that span flat as float4
 * (lane = consecutive 16 bytes: fully coalesced, non-temporal), every load of a thread in flight before the first use,
 * and leave the four squared differences of each float4 in LDS; thread t then adds the G squares of item t (stride G) and
 * stores one reward and one flag, contiguously.  One workgroup per 256 items, no grid-stride below 2^20 workgroups.
 * Measured (tools/reward_flat_variants.hip, 16 Mi pairs, G = 7 / 9 / 12 / 13 / 16 / 19): 6.0-6.1 TB/s at every G against
 * 5.2-5.6 for round 3's kernel (dword loads when G % 4 != 0, loads issued one per loop trip, 8192 workgroups striding);
 * partial sums per float4 when G % 4 == 0 (less LDS traffic) measured 5.9, flags packed into dwords the same as bytes. */
__global__ void __launch_bounds__(256) pmg_k_reward_flat(const float* __restrict__ ag, const float* __restrict__ dg, long long B, int G,
                                                        float thr, int binary, float* __restrict__ reward,
                                                        unsigned char* __restrict__ ok)
{
    constexpr int MAXQ = 5;                                    /* float4 per thread per array: G <= 20 */
    __shared__ float4 sq[MAXQ * 256];
    const int t = (int)threadIdx.x;
    const int q4 = G * 64;                                     /* float4 per array of a full workgroup */
    for (long long base = (long long)blockIdx.x * 256; base + 256 <= B; base += (long long)gridDim.x * 256) {
        const float4* a = (const float4*)(ag + base * G);
        const float4* d = (const float4*)(dg + base * G);
        float4 x[MAXQ], y[MAXQ];
#pragma unroll
        for (int k = 0; k < MAXQ; k++) {
            const int w = t + 256 * k;
            if (w < q4) { x[k] = nt::load4(a + w); y[k] = nt::load4(d + w); }
        }
#pragma unroll
        for (int k = 0; k < MAXQ; k++) {
            const int w = t + 256 * k;
            if (w < q4) {
                const float e0 = x[k].x - y[k].x, e1 = x[k].y - y[k].y, e2 = x[k].z - y[k].z, e3 = x[k].w - y[k].w;
                sq[w] = make_float4(e0 * e0, e1 * e1, e2 * e2, e3 * e3);
            }
        }
        __syncthreads();
        const float* part = (const float*)sq + t * G;
        float s = 0.f;
        for (int k = 0; k < G; k++) s += part[k];
        const float dist = sqrtf(s);
        const bool na = dist > thr;
        if (reward) reward[base + t] = binary ? (na ? -1.f : -0.f) : -dist;
        if (ok) ok[base + t] = na ? 0 : 1;
        __syncthreads();
    }
}
/* any G, and the < 4 tail items of the G == 3 path */
__global__ void __launch_bounds__(256) pmg_k_reward(const float* __restrict__ ag, const float* __restrict__ dg, long long first,
                                                   long long B, int G, float thr, int binary, float* __restrict__ reward,
                                                   unsigned char* __restrict__ ok)
{
    long long i = first + (long long)blockIdx.x * blockDim.x + threadIdx.x;
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
    /* One workgroup plans a small batch in one launch.  From 4096 envs on the two-pass plan over ceil(N / 1024) workgroups
     * takes over (same lists, tests/test_emulated_kernels.py): every thread classifies ONE env, so the two launches together
     * are shorter than the single workgroup's walk over the batch in chunks (13 us at 4096 envs -- reach + 0.7 %,
     * pick_and_place x 8192 + 0.8 %, push + 0.4 % measured --, 32 us with several blocks, 0.9 ms at 65 536).
     * PMG_PLAN_TWO_PASS=0 / 1 forces either (experiments; batches beyond PLAN_SINGLE_MAX always take two passes) */
    static const int force_two_pass = getenv("PMG_PLAN_TWO_PASS") ? atoi(getenv("PMG_PLAN_TWO_PASS")) : -1;
    const bool two_pass = P.n_envs > pmg::PLAN_SINGLE_MAX || (force_two_pass >= 0 ? force_two_pass != 0 : P.n_envs >= pmg::PLAN_TWO_PASS_MIN);
    if (!two_pass && P.nb == 0 && !P.joint_control) hipLaunchKernelGGL(pmg_k_plan_reach, dim3(1), dim3(pmg::PLAN_THREADS), 0, s, P, d_actions);
    else if (!two_pass) hipLaunchKernelGGL(pmg_k_plan, dim3(1), dim3(pmg::PLAN_THREADS), 0, s, P, d_actions);
    else {
        const int nwg = (P.n_envs + pmg::PLAN_THREADS - 1) / pmg::PLAN_THREADS;
        hipLaunchKernelGGL(pmg_k_plan_count, dim3(nwg), dim3(pmg::PLAN_THREADS), 0, s, P, d_actions);
        hipLaunchKernelGGL(pmg_k_plan_scatter, dim3(nwg), dim3(pmg::PLAN_THREADS), 0, s, P, d_actions);
    }
    return hipGetLastError();
}
/* one free object: the fast-path list four envs per wavefront (pmg_packed.h; 31 KB of LDS per workgroup).  The envs
 * whose gripper works on the object run one per wavefront with the full 24-contact store in a SEPARATE launch
 * (pmg_k_step_list<1,24,0,CYL>, 16 KB) on the side stream, so that a batch in which most envs are of that kind keeps
 * the occupancy it had before the packing; surplus workgroups at the end of either grid exit at once */
/* (A helper wavefront for the packed kernel, too -- the narrowphase of the four envs beside their dynamics -- measured SLOWER in round 3:
 * push 1.39 -> 1.10 M, slide 1.23 -> 0.95 M, pick_and_place 1.86 -> 1.25 M: the 1024 packed wavefronts of a 4096-env step sit one per SIMD,
 * and a second wavefront on every SIMD costs these kernels more than the 5.6 k cycles per substep it takes off the chain.  Removed.) */
constexpr int OBJ4_THREADS = 64;
template <bool CYL>
__global__ void __launch_bounds__(OBJ4_THREADS, PMG_WAVES_PER_EU) pmg_k_step_obj4(pmg::EnvParams P, const float* __restrict__ actions)
{
    __shared__ pmgp::ObjLds4 sm;
    pmgp::step_group_obj<CYL>(P, actions, (int)blockIdx.x, sm);
}
template <bool CYL>
__global__ void __launch_bounds__(64, PMG_WAVES_PER_EU) pmg_k_redo_obj(pmg::EnvParams P, const float* __restrict__ actions)
{
    const int* redo = P.sched + 2 + 2 * P.n_envs;
    if ((int)blockIdx.x >= redo[0]) return;
    pmg::step_env<1, 24, CYL>(P, actions, redo[1 + blockIdx.x]);
}
/* several free blocks (block_stack / block_rearrange): list 0 = envs whose gripper works on a block, with the full
 * 48-contact store; list 1 = the rest with a 30-contact store (20 KB of LDS instead of 29: 8 workgroups per CU instead
 * of 5).  The two launches run concurrently on two streams; a list-1 env that overflows is queued for the redo pass */
#ifndef PMG_LIST_TWO_WAVES
#define PMG_LIST_TWO_WAVES 1
#endif
constexpr int MULTI_SMALL_MAXC = 30;
constexpr int LIST0_THREADS = PMG_LIST_TWO_WAVES ? 128 : 64;
/* LIST 0 (the full contact store: the envs whose gripper works on an object -- the long pole of a batched step) runs
 * with TWO wavefronts per workgroup: the second one collides while the first computes the dynamics (helper_wave_loop) --
 * except on the lid task (chest_pick_and_place, CYL == 3): a quarter of its batch is on list 0, the step is bound by the
 * wavefront slots, not by one chain, and the helper wavefronts cost more slots than they shorten chains (0.580 -> 0.609 M
 * without them, round 4; with them 0.614 -> 0.585 M, round 5; chest_push neutral, the block tasks lose 1 % without: they keep theirs) */
constexpr bool list_two_waves(int list, int cyl) { return list == 0 && PMG_LIST_TWO_WAVES != 0 && cyl != 3; }
/* ... and slide's list 0 with a THIRD one, which repeats the finger x puck pairs in double beside the helper's float narrowphase
 * (pmg::SpecLds; the rule that brings slide's single steps to the chaos floor, off the critical path) */
/* (The chest tasks' cylinder repeats stay SERIAL: their steps are bound by wavefront slots, not by one chain -- a third wavefront on
 * their list 0, with the plan sending every env whose gripper base can reach the chest there, measured 0.463 -> 0.263 M on chest_push,
 * the lid task with helper + third wavefront 0.610 -> 0.366 M; profiles/r06_chest_repeat_rule_experiment.txt) */
constexpr bool list_spec_wave(int list, int cyl) { return list_two_waves(list, cyl) && cyl == 1 && PMG_CYL_SPEC != 0 && PMG_CYL_PUSH_ALL != 0 && PMG_CYL_REDO64 != 0; }
constexpr int list_threads(int list, int cyl) { return list_spec_wave(list, cyl) ? 192 : (list_two_waves(list, cyl) ? 128 : 64); }
template <bool ON, int CYL> struct SpecStore { __device__ static __forceinline__ pmg::SpecLds<CYL>* get() { return nullptr; } };
template <int CYL> struct SpecStore<true, CYL> { __device__ static __forceinline__ pmg::SpecLds<CYL>* get() { __shared__ pmg::SpecLds<CYL> s; return &s; } };
template <int NB, int MAXC, int LIST, int CYL = 0>
__global__ void __launch_bounds__(list_threads(LIST, CYL), PMG_WAVES_PER_EU) pmg_k_step_list(pmg::EnvParams P, const float* __restrict__ actions)
{
    __shared__ pmg::ContactLds<NB, MAXC> L;
    __shared__ pmg::LaneTabStore lcs;
    pmg::SpecLds<CYL>* sp = SpecStore<list_spec_wave(LIST, CYL), CYL>::get();
    const int b = (int)blockIdx.x;
    if (b >= P.sched[LIST]) return;
    /* issue priority for the wavefronts that are the long pole of the step: list 0 of the multi-block / chest tasks
     * (block_stack-4 +2.6 %), list 0 of a one-object task when the plan moved the fingers-down class there
     * (pick_and_place 1.59 -> 1.85 M; push / slide, whose long pole is the packed fingers-down wavefront, lose 3 / 8 %
     * with it and do not promote); PMG_LIST0_PRIO overrides (tools/prio_exp.sh) */
    /* ... and slide's three-wavefront list 0 (round 6): with the double repeat beside its narrowphase the pushing env is the step's
     * longest chain: 1.208 -> 1.240 M with priority (PMG_LIST0_PRIO=0 / 1, same library) */
    if (LIST == 0) wv::set_priority(P.list0_prio >= 0 ? P.list0_prio : ((NB > 1 || list_spec_wave(LIST, CYL) || *pmg::plan_promoted(P)) ? 1 : 0));
    const int env = P.sched[2 + LIST * P.n_envs + b];
    const bool ok = pmg::step_env_core<NB, MAXC, CYL, list_two_waves(LIST, CYL), list_spec_wave(LIST, CYL)>(P, actions, env, L, lcs, true, sp);
    if (!ok && threadIdx.x == 0) {
        int* redo = P.sched + 2 + 2 * P.n_envs;
        int slot = atomicAdd(redo, 1);
        redo[1 + slot] = env;
    }
}
__global__ void __launch_bounds__(64, PMG_WAVES_PER_EU) pmg_k_redo_multi(pmg::EnvParams P, const float* __restrict__ actions)
{
    const int* redo = P.sched + 2 + 2 * P.n_envs;
    if ((int)blockIdx.x >= redo[0]) return;
    pmg::step_env<5, 48, false>(P, actions, redo[1 + blockIdx.x]);
}
/* chest tasks: the same two-list split -- list 0 (gripper at the chest or at a block) keeps the full layout (48 contacts,
 * a stage slot per pair, 32 000 B = 5 workgroups per CU), list 1 runs ContactLds<6, 30> with ranked stage slots
 * (CHEST_SMALL); an env of list 1 whose contacts or surviving pairs overflow is recomputed by pmg_k_redo_chest */
template <int CYL>
__global__ void __launch_bounds__(64, PMG_WAVES_PER_EU) pmg_k_redo_chest(pmg::EnvParams P, const float* __restrict__ actions)
{
    const int* redo = P.sched + 2 + 2 * P.n_envs;
    if ((int)blockIdx.x >= redo[0]) return;
    pmg::step_env<6, 48, CYL>(P, actions, redo[1 + blockIdx.x]);
}
/* (a redo list walked with a stride by 1024 workgroups instead of one workgroup per env leaving on its first load would save the
 * dispatch of 4096 empty workgroups, ~3 us -- but step_env inside that loop spills ~100 VGPRs to scratch, and as a noinline
 * function it needs a 500-byte frame: measured, not kept) */
static inline int redo_grid(int n_envs) { return n_envs; }
/* every runtime call of the fork / join between the step's two concurrent launches is checked: a failed event record or stream
 * wait would silently serialise the two lists, or let the redo pass race them -- the error goes back to the caller
 * (pmg_step*: PMG_ERR_HIP + pmg_last_error) */
#define PMG_FJ(call) do { hipError_t fj_e_ = (call); if (fj_e_ != hipSuccess) return fj_e_; } while (0)
hipError_t pmg_launch_step(const pmg::EnvParams& P, const float* d_actions, hipStream_t s, int packed, hipStream_t side,
                           hipEvent_t ev_fork, hipEvent_t ev_join)
{
    /* Which of the two concurrent launches is submitted first.  list0_first: the full-store list -- the step's long
     * wavefronts -- goes to the main stream at once and the fast-path list follows on the side stream behind the fork event
     * (a few us later), so the long chains are placed on an empty machine.  Submitted second they started behind the
     * first ROUND of the fast-path list's workgroups (3800 one-env wavefronts on 2048 slots: 1.8 ms late on a 4 ms chain):
     * block_stack-4 0.66 -> 0.79 M, block_rearrange-4 0.54 -> 0.70 M, chest_pick_and_place-4 0.45 -> 0.57 M, chest_push-4
     * 0.40 -> 0.42 M.  With ONE object the packed launch is the pole itself and stays first (push 1.63 -> 1.48 M otherwise) --
     * until its workgroups alone fill the machine: five of them (31.8 KB of LDS each) hold a compute unit's LDS, so beyond
     * 5 x CUs packed workgroups (wave_budget = 6 x CUs; 5120 envs) the one-env list waited for the first round of the packed
     * launch to drain: pick_and_place x 8192 2.64 -> 2.84 M, x 6144 1.99 -> 2.13 M, push x 6144 1.74 -> 1.94 M (push / slide
     * x 8192 and pick_and_place x 16 384: neutral).
     * PMG_LIST0_FIRST=0 / 1 forces either order for every task (experiments) */
    static const int force_first = getenv("PMG_LIST0_FIRST") ? atoi(getenv("PMG_LIST0_FIRST")) : -1;
    const bool list0_first = force_first >= 0 ? force_first != 0 : (P.nb > 1 || P.chest >= 0 || (P.n_envs + 3) / 4 > P.wave_budget * 5 / 6);
    if (P.chest >= 0 && packed) {
        PMG_FJ(hipEventRecord(ev_fork, s));
        PMG_FJ(hipStreamWaitEvent(side, ev_fork, 0));
        hipStream_t s0 = list0_first ? s : side, s1 = list0_first ? side : s;
        if (P.chest == 0) hipLaunchKernelGGL((pmg_k_step_list<6, 48, 0, 2>), dim3(P.n_envs), dim3(list_threads(0, 2)), 0, s0, P, d_actions);
        else hipLaunchKernelGGL((pmg_k_step_list<6, 48, 0, 3>), dim3(P.n_envs), dim3(list_threads(0, 3)), 0, s0, P, d_actions);
        if (!list0_first) PMG_FJ(hipEventRecord(ev_join, side));
        if (P.chest == 0) hipLaunchKernelGGL((pmg_k_step_list<6, MULTI_SMALL_MAXC, 1, 2>), dim3(P.n_envs), dim3(64), 0, s1, P, d_actions);
        else hipLaunchKernelGGL((pmg_k_step_list<6, MULTI_SMALL_MAXC, 1, 3>), dim3(P.n_envs), dim3(64), 0, s1, P, d_actions);
        if (list0_first) PMG_FJ(hipEventRecord(ev_join, side));
        PMG_FJ(hipStreamWaitEvent(s, ev_join, 0));
        if (P.chest == 0) hipLaunchKernelGGL((pmg_k_redo_chest<2>), dim3(redo_grid(P.n_envs)), dim3(64), 0, s, P, d_actions);
        else hipLaunchKernelGGL((pmg_k_redo_chest<3>), dim3(redo_grid(P.n_envs)), dim3(64), 0, s, P, d_actions);
        return hipGetLastError();
    }
    if (P.chest >= 0) {
        /* chest tasks: one env per wavefront with the chest layout (door slot + chest pairs, 47 KB of LDS) */
        if (P.chest == 0) hipLaunchKernelGGL((pmg_k_step<6, 48, 2>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
        else hipLaunchKernelGGL((pmg_k_step<6, 48, 3>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
        return hipGetLastError();
    }
    if (P.nb > 1 && packed) {
        PMG_FJ(hipEventRecord(ev_fork, s));
        PMG_FJ(hipStreamWaitEvent(side, ev_fork, 0));
        hipStream_t s0 = list0_first ? s : side, s1 = list0_first ? side : s;
        hipLaunchKernelGGL((pmg_k_step_list<5, 48, 0>), dim3(P.n_envs), dim3(LIST0_THREADS), 0, s0, P, d_actions);
        if (!list0_first) PMG_FJ(hipEventRecord(ev_join, side));
        /* up to four blocks: 24 candidate pairs instead of 32 keep the narrowphase workspace under the row store (20 KB) */
        if (P.nb <= 4) hipLaunchKernelGGL((pmg_k_step_list<4, MULTI_SMALL_MAXC, 1>), dim3(P.n_envs), dim3(64), 0, s1, P, d_actions);
        else hipLaunchKernelGGL((pmg_k_step_list<5, MULTI_SMALL_MAXC, 1>), dim3(P.n_envs), dim3(64), 0, s1, P, d_actions);
        if (list0_first) PMG_FJ(hipEventRecord(ev_join, side));
        PMG_FJ(hipStreamWaitEvent(s, ev_join, 0));
        hipLaunchKernelGGL(pmg_k_redo_multi, dim3(redo_grid(P.n_envs)), dim3(64), 0, s, P, d_actions);
        return hipGetLastError();
    }
    if (P.nb == 1 && packed) {
        PMG_FJ(hipEventRecord(ev_fork, s));
        PMG_FJ(hipStreamWaitEvent(side, ev_fork, 0));
        const int groups = (P.n_envs + 3) / 4;
        hipStream_t s0 = list0_first ? s : side, s1 = list0_first ? side : s;
        if (P.task == PMG_TASK_SLIDE) {
            hipLaunchKernelGGL((pmg_k_step_list<1, 24, 0, true>), dim3(P.n_envs), dim3(list_threads(0, 1)), 0, s0, P, d_actions);
            if (!list0_first) PMG_FJ(hipEventRecord(ev_join, side));
            hipLaunchKernelGGL((pmg_k_step_obj4<true>), dim3(groups), dim3(OBJ4_THREADS), 0, s1, P, d_actions);
            if (list0_first) PMG_FJ(hipEventRecord(ev_join, side));
            PMG_FJ(hipStreamWaitEvent(s, ev_join, 0));
            hipLaunchKernelGGL((pmg_k_redo_obj<true>), dim3(redo_grid(P.n_envs)), dim3(64), 0, s, P, d_actions);
        } else {
            hipLaunchKernelGGL((pmg_k_step_list<1, 24, 0, false>), dim3(P.n_envs), dim3(LIST0_THREADS), 0, s0, P, d_actions);
            if (!list0_first) PMG_FJ(hipEventRecord(ev_join, side));
            hipLaunchKernelGGL((pmg_k_step_obj4<false>), dim3(groups), dim3(OBJ4_THREADS), 0, s1, P, d_actions);
            if (list0_first) PMG_FJ(hipEventRecord(ev_join, side));
            PMG_FJ(hipStreamWaitEvent(s, ev_join, 0));
            hipLaunchKernelGGL((pmg_k_redo_obj<false>), dim3(redo_grid(P.n_envs)), dim3(64), 0, s, P, d_actions);
        }
        return hipGetLastError();
    }
    if (P.nb == 0 && packed) {
        /* the one-wavefront kernel FIRST: when it is its turn (a contact-free step) its 1024 wavefronts are placed on an
         * empty machine, one per SIMD; behind the other kernel's draining empty workgroups they were not (0.82 ms) */
        /* n_prone + ceil(n_free / 4) <= N workgroups; deferring (packed == 2) it runs on contact-free steps only: ceil(N / 4) */
        hipLaunchKernelGGL(pmg_k_step_reach, dim3(packed == 2 ? (P.n_envs + 3) / 4 : P.n_envs), dim3(64), 0, s, P, d_actions, packed == 2 ? 1 : 0);
        if (packed == 2) hipLaunchKernelGGL(pmg_k_step_reach2, dim3(P.n_envs), dim3(128), 0, s, P, d_actions);   /* n_prone + ceil(n_free / 8) <= N */
        /* mispredictions are rare; surplus workgroups of the redo grid exit on their first instruction */
        hipLaunchKernelGGL(pmg_k_redo, dim3(redo_grid(P.n_envs)), dim3(64), 0, s, P, d_actions);
    } else if (P.nb == 0) hipLaunchKernelGGL((pmg_k_step<0, 8, false>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
    else if (P.task == PMG_TASK_SLIDE) hipLaunchKernelGGL((pmg_k_step<1, 24, true>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
    else if (P.nb == 1) hipLaunchKernelGGL((pmg_k_step<1, 24, false>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
    else hipLaunchKernelGGL((pmg_k_step<5, 48, false>), dim3(P.n_envs), dim3(64), 0, s, P, d_actions);
    return hipGetLastError();
}
/* set_sub_goal (kuka_multi_step_base_env.py:154-177): new level for the masked envs, desired_goal refreshed in
 * the packed output rows; 32 threads per env: goal components (<= 19) + one for the level */
__global__ void __launch_bounds__(256) pmg_k_sub_goal(pmg::EnvParams P, const unsigned char* __restrict__ mask, int level)
{
    int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int env = t / 32, i = t % 32;
    if (env >= P.n_envs || (mask != nullptr && mask[env] == 0)) return;
    float v = 0.f;
    float* cold = P.cold + (size_t)env * pmg::COLD_DIM;
    if (i == 31) { cold[7] = (float)level; return; }
    if (i >= P.gdim) return;
    v = pmg::effective_goal_at_level(P, env, i, level); /* level passed in, not read back: thread 31 may not have stored it yet */
    /* chest sub-goal 0: the gripper goal is the gripper's current pose = the tail of the achieved goal of the last observation */
    if (P.chest >= 0 && P.grip_goal && level == 0 && i >= 1 + 3 * P.nb) v = P.out[(size_t)env * P.packed + P.odim + P.pdim + i];
    P.out[(size_t)env * P.packed + P.odim + P.pdim + P.gdim + i] = v;
}
hipError_t pmg_launch_sub_goal(const pmg::EnvParams& P, const unsigned char* d_mask, int level, hipStream_t s)
{
    int threads = P.n_envs * 32;
    hipLaunchKernelGGL(pmg_k_sub_goal, dim3((threads + 255) / 256), dim3(256), 0, s, P, d_mask, level);
    return hipGetLastError();
}
hipError_t pmg_launch_reset(const pmg::EnvParams& P, const unsigned char* d_mask, hipStream_t s, int done_only)
{
    hipLaunchKernelGGL(pmg_k_reset, dim3(P.n_envs), dim3(64), 0, s, P, d_mask, done_only);
    return hipGetLastError();
}
hipError_t pmg_launch_reward(const float* ag, const float* dg, long long B, int G, float thr, int binary, float* reward,
                             unsigned char* ok, hipStream_t s)
{
    if (B <= 0) return hipSuccess;
    long long first = 0;
    bool aligned = ((((size_t)ag | (size_t)dg | (size_t)reward) & 15) == 0) && (((size_t)ok & 3) == 0);
    if (G == 3 && aligned && B >= 4) {
        long long quads = B / 4;
        long long want = (quads + 255) / 256;
        unsigned grid = (unsigned)(want < 65536 ? want : 65536); /* a workgroup per 256 quads (measured best), grid-stride beyond 64 Mi items */
        hipLaunchKernelGGL(pmg_k_reward3, dim3(grid), dim3(256), 0, s, (const float4*)ag, (const float4*)dg, quads, thr, binary,
                           (float4*)reward, (unsigned int*)ok);
        first = quads * 4;
    }
    static const int generic_only = getenv("PMG_REWARD_GENERIC") ? atoi(getenv("PMG_REWARD_GENERIC")) : 0;   /* (counter calibration: the dword kernel) */
    if (G > 3 && G <= 20 && B >= 256 && ((((size_t)ag | (size_t)dg) & 15) == 0) && !generic_only) {
        long long want = B / 256;                                /* full workgroups; the < 256 tail items go to pmg_k_reward */
        unsigned grid = (unsigned)(want < (1 << 20) ? want : (1 << 20));
        hipLaunchKernelGGL(pmg_k_reward_flat, dim3(grid), dim3(256), 0, s, ag, dg, B, G, thr, binary, reward, ok);
        first = want * 256;
    }
    if (first < B) {
        unsigned grid = (unsigned)((B - first + 255) / 256);
        hipLaunchKernelGGL(pmg_k_reward, dim3(grid), dim3(256), 0, s, ag, dg, first, B, G, thr, binary, reward, ok);
    }
    return hipGetLastError();
}
