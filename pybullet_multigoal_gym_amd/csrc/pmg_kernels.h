/*
 * pmg_kernels.h -- per-environment step / reset bodies (one wavefront each).
 * Included by pmg_kernels.hip (the gfx950 build) and by the test-only
 * emulator translation unit.
 */
#ifndef PMG_KERNELS_H
#define PMG_KERNELS_H

#include "pmg_contact.h"

#ifndef PMG_COLD_CONTACTS
#define PMG_COLD_CONTACTS 1 /* keep the reach kernel's rare contact phases out of line (compact hot loop) */
#endif

/* kernel parameters: a namespace of its own, so that argument-dependent lookup never mixes the two lane-layout
 * namespaces (pmg / pmgp) that both take it */
namespace pmgx {
struct EnvParams {
    int n_envs, task, nb, grasping, has_obj, joint_control, binary_reward, max_steps, in_air, random_order;
    int multi;              /* multi-block observation layout: block_stack / block_rearrange */
    int curriculum, curriculum_update; /* kuka_multi_step_base_env.py:121-152 */
    int decomposition, grip_goal;      /* task_decomposition, grip_informed_goal (goal = blocks | tip target | finger width) */
    int chest;                         /* -1, or the chest of chest_push (0: front sliding door) / chest_pick_and_place (1: lid);
                                          the door's joint position, velocity and motor latch live in goal[0..2] */
    double goals_per_curriculum;
    int adim, odim, pdim, gdim, packed;
    float chest_reach;      /* plan: how far in front of the chest's front face a tip target counts as "at the chest" */
    float near_r;           /* plan: a tip target within this distance of a free object sends the env to the full-store list */
    int wave_budget;        /* 1.5 wavefronts per SIMD of THIS device (6 x CUs: 1536 on an MI355X): pmg_k_plan's promotion rule */
    int fd_div;             /* plan: the fingers-down class of a one-object task goes to list 0: -1 always (pick_and_place: the default), 0 never (push,
                               slide) -- a rule of the TASK, so that an env's kernel, and with it the float32 order of its sums, does not depend on
                               the batch it is in; > 0 (experiments, PMG_FD_DIV): while the class is under 1 / fd_div of the batch and the step fits
                               the wavefront budget, round 2-4's rule */
    int list0_prio;         /* s_setprio level of the full-store (list 0) wavefronts; -1: when they are the long pole of the step (pmg_k_step_list) */
    float thr;
    float ee_lo[3], ee_hi[3];
    float table_c[3], table_h[3], table_mu;
    /* sampling boxes, kept in double so that the RNG draws reproduce numpy's float64 uniform() */
    double tip_init[3], obj_lo[3], obj_hi[3], tgt_lo[3], tgt_hi[3], obj_z;
    /* device arrays */
    float* hot;      /* [N, HOT_DIM]  */
    float* cold;     /* [N, COLD_DIM] */
    float* goal;     /* [N, GOAL_DIM]: static targets (stack: per block; rearrange: per target slot), [15] = moved mask */
    float* curr;     /* [N, CURR_DIM] curriculum state: prob[5] generated[5] goal_step (NULL without use_curriculum) */
    float* blocks;   /* [N, BLOCK_DIM * nb] */
    unsigned* rng;   /* [N, 625] MT19937 state + index */
    float* out;      /* [N, packed]: obs | policy | ag | dg | reward | goal_achieved | done */
    int* sched;      /* [4 + 3N + 3 ceil(N / 1024)]: [0] [1] counts of {contact-prone, other} envs, [2 .. 2 + 2N) the two env
                        lists, [2 + 2N] the redo count and behind it the redo list of the fast paths (pmg_packed.h); then
                        [3 + 3N ..) the per-workgroup class counts of the two-pass plan (3 per 1024 envs) and one last word:
                        did the plan promote the fingers-down class to list 0 (plan_promoted()) */
    int lpt_thresh;  /* > 0 (and env_cycles kept): several blocks, fast-path list ordered longest-first -- envs whose wavefront took more than this many cycles / 64 in the PREVIOUS step lead the list (plan_class).  A constant only as an override (PMG_LPT_CYCLES) */
    int lpt_permille; /* > 0: the threshold is DERIVED ON THE DEVICE -- the cycle count above which the slowest lpt_permille / 1000 of the fast-path list's envs lay in the last step (a 128-bin histogram the plan keeps: lpt_state): no per-part, per-batch-size tuned constants */
    int lpt_parity;   /* which of the two threshold words this step's plan reads (it writes the other one for the next step) */
    int* lpt_state;   /* [2 + LPT_BINS]: thresholds (two, by step parity) and the histogram of env_cycles over the fast-path envs */
    int* env_cycles; /* diagnostics (NULL unless PMG_ENV_CYCLES=1 at creation): [N, 2] shader cycles / 64 the env's wavefront spent
                        in its last step, and the largest contact count any of its substeps saw */
#ifdef PMG_PROFILE
    long long* prof; /* [32] per-phase shader cycles of env 0 */
#endif
};
}  // namespace pmgx

namespace pmg {

#ifdef PMG_PROFILE
#define PMG_TICK(i) PMG_STAMP(tprev, i)
#else
#define PMG_TICK(i) do { } while (0)
#endif

using pmgx::EnvParams;

}  // namespace pmg
#define WV wv
namespace pmg {
#include "pmg_step_body.inc"
}  // namespace pmg
#undef WV
#define WV wr
namespace pmgp {
using pmgx::EnvParams;
#include "pmg_step_body.inc"
}  // namespace pmgp
#undef WV
namespace pmg {

/* one env per wavefront: the workgroup's LDS holds this env's contact store and the lane-constant table */
template <int NB, int MAXC, int CYL, bool TWO = false>
__device__ __forceinline__ void step_env(const EnvParams& P, const float* actions, int env)
{
    __shared__ ContactLds<NB, MAXC> L;
    __shared__ LaneTabStore lcs;
    if (env >= P.n_envs) return;
    step_env_core<NB, MAXC, CYL, TWO>(P, actions, env, L, lcs, true);
}

/* ------------------------------------------------------------------ */
/* Launch-order plan.  A batched step lasts as long as its slowest wavefront, and the slow ones are
 * the envs whose fingers will touch the table (contact phases, ~2.5x a contact-free substep).  The
 * hardware favours the OLDEST wave of a SIMD, so those envs are given the lowest workgroup ids:
 * dispatched first, one per SIMD, they run at single-wave speed from t = 0 while the rest fill the
 * issue slots.  The mapping never changes a result (envs are independent), only who waits.     */
/* forward kinematics of ONE env by ONE thread (the plan kernel's view of an env): lowest point of either finger box
 * and the tip position, from the joint angles -- same chain as fk() of pmg_device_body.inc, without the wave */
__device__ inline void plan_fk(const float* q, float& finger_z, float* tip)
{
    float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}, p[3] = {0.f, 0.f, 0.f};
    float R6[9], p6[3];
    const float fh[3] = PMG_FINGER_HALF;
    finger_z = 1e30f;
    for (int j = 0; j < NJ; j++) {
        if (j == 7) { for (int a = 0; a < 9; a++) R6[a] = R[a]; for (int a = 0; a < 3; a++) p6[a] = p[a]; }
        if (j == 8) { for (int a = 0; a < 9; a++) R[a] = R6[a]; for (int a = 0; a < 3; a++) p[a] = p6[a]; } /* finger 2 hangs off link 7 too */
        float L[9], o[3];
        if (C_JTYPE[j] != 0) {            /* prismatic along the joint axis */
            for (int a = 0; a < 9; a++) L[a] = C_JROT[j][a / 3][a % 3];
            float d[3];
            for (int r = 0; r < 3; r++) d[r] = L[3 * r] * C_JAXIS[j][0] + L[3 * r + 1] * C_JAXIS[j][1] + L[3 * r + 2] * C_JAXIS[j][2];
            for (int a = 0; a < 3; a++) o[a] = C_JXYZ[j][a] + d[a] * q[j];
        } else {                          /* revolute about local z */
            float s, c;
            sincosf(q[j], &s, &c);
            for (int r = 0; r < 3; r++) {
                float rx = C_JROT[j][r][0], ry = C_JROT[j][r][1];
                L[3 * r] = rx * c + ry * s; L[3 * r + 1] = ry * c - rx * s; L[3 * r + 2] = C_JROT[j][r][2];
                o[r] = C_JXYZ[j][r];
            }
        }
        float t[3];
        mat3v(R, o, t);
        p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
        mat3m(R, L, R);
        if (j >= 7) finger_z = fminf(finger_z, p[2] - (fabsf(R[6]) * fh[0] + fabsf(R[7]) * fh[1] + fabsf(R[8]) * fh[2]));
    }
    for (int a = 0; a < 3; a++) tip[a] = p6[a] + R6[3 * a + 2] * TIP_Z;
}

/* what every tip-control classification reads: the env's tip target and the first three action components.  Fetched
 * unconditionally and up front, so that the loads of several envs of one thread are in flight together (plan_all) */
struct PlanIn { float ee[3], a[3]; };
template <bool Z_ONLY = false>
__device__ __forceinline__ PlanIn plan_fetch(const EnvParams& P, const float* actions, int env)
{
    PlanIn in;
    const float* hot = P.hot + (size_t)env * HOT_DIM;
    const float* act = actions + (size_t)env * P.adim;
    /* Z_ONLY (reach): two loads per env instead of six.  The plan is ONE workgroup on one compute unit and these are
     * per-lane cache lines: 1024 threads x 4 chunks x 6 loads kept its texture-address unit busy for ~5 us (timestamps
     * inside the kernel: fetch + first barrier 5.5 of its 8.6 us) */
#pragma unroll
    for (int a = Z_ONLY ? 2 : 0; a < 3; a++) { in.ee[a] = hot[18 + a]; in.a[a] = act[a]; }   /* (every task's action has >= 3 components) */
    return in;
}
__device__ __forceinline__ bool contact_prone(const EnvParams& P, const float* actions, int env, const PlanIn& in)
{
    const float* hot = P.hot + (size_t)env * HOT_DIM;
    if (P.joint_control) {
        /* joint targets move by up to 0.05 rad per step (kuka.py:205): with the ~0.8 m reach of the arm a finger or the
         * tip travels at most ~4.5 cm.  Prone = a finger could get inside the contact margin of the table, or the tip
         * within reach of a free object (6.5 cm, as under tip control) */
        float q[NJ], fz, tip[3];
        for (int d = 0; d < NJ; d++) q[d] = hot[d];
        plan_fk(q, fz, tip);
        if (fz < P.table_c[2] + P.table_h[2] + 0.045f + CONTACT_MARGIN) return true;
        if (P.nb > 1) return true;                       /* several blocks: keep the full store under joint control */
        for (int b = 0; b < P.nb; b++) {
            const float* pb = P.blocks + ((size_t)env * P.nb + b) * BLOCK_DIM;
            float d2 = (tip[0] - pb[0]) * (tip[0] - pb[0]) + (tip[1] - pb[1]) * (tip[1] - pb[1]) + (tip[2] - pb[2]) * (tip[2] - pb[2]);
            if (d2 < 0.11f * 0.11f) return true;
        }
        return false;
    }
    if (P.nb >= 1) {
        /* free objects: the fast paths store fewer contacts per env than the full kernels (pmg_packed.h,
         * pmg_k_step_list); the count only grows past that when the fingers work on an object, i.e. when the tip
         * target comes within 6.5 cm of one */
        float t[3];
        for (int a = 0; a < 3; a++) t[a] = fminf(fmaxf(in.ee[a] + in.a[a] * 0.01f, P.ee_lo[a]), P.ee_hi[a]);
        bool near = false;
        for (int b = 0; b < P.nb; b++) {
            const float* p = P.blocks + ((size_t)env * P.nb + b) * BLOCK_DIM;
            float d2 = (t[0] - p[0]) * (t[0] - p[0]) + (t[1] - p[1]) * (t[1] - p[1]) + (t[2] - p[2]) * (t[2] - p[2]);
            near = near || d2 < P.near_r * P.near_r;
        }
        if (P.chest >= 0) {
            /* the chest: its walls / door / lid / handle add contacts once the fingers can TOUCH them: the finger boxes
             * (1.25 cm from the tool axis in x, 4.5 cm in y) against the front face at x = -0.592 (door) / -0.555 (the
             * lid's handle), `chest_reach` = that + this step's target motion; the door slides to y = +0.19, the lid to
             * x = -0.805.  (The gripper-base cylinder grazes the door's top edge earlier, with <= 4 more contacts: they
             * fit.)  A prediction only: an env of the fast-path list whose contacts do not fit its store is recomputed by
             * the redo pass */
            const float x0 = P.chest == 0 ? -0.705f : -0.805f, x1 = P.chest == 0 ? -0.592f : -0.555f, y1 = P.chest == 0 ? 0.19f : 0.07f;
            near = near || (t[0] > x0 - 0.064f && t[0] < x1 + P.chest_reach && t[1] > -0.07f - 0.064f && t[1] < y1 + 0.064f && t[2] < 0.272f + 0.015f);
        }
        return near;
    }
    float z = in.ee[2];                                   /* tip target: the tip is within mm of it */
    float zn = fminf(fmaxf(z + in.a[2] * 0.01f, P.ee_lo[2]), P.ee_hi[2]);
    return fminf(z, zn) < P.ee_lo[2] + 0.012f;
}
/* one 1024-thread workgroup partitions all envs (stable, no atomics): per-wave ballots, counts of
 * every (chunk, wave) tile in LDS, then each tile scatters at its exclusive prefix */
constexpr int PLAN_THREADS = 1024, PLAN_MAX_TILES = 1024;
constexpr int PLAN_SINGLE_MAX = 16384;   /* the single-workgroup plan serves batches up to here (<= PLAN_MAX_TILES * 64) */
constexpr int PLAN_TWO_PASS_MIN = 4096;  /* ... and by default only below here: from four workgroups on the two-pass plan is the shorter one (pmg_launch_plan) */
#ifndef PMG_FD_DIV
#define PMG_FD_DIV 8
#endif
/* class of an env for the launch order: 0 = first list (one env per wavefront, full contact store), 1 / 2 = second
 * list (fast path).  With free objects the fast-path envs are grouped by whether the fingers are down at the table
 * (class 1) or up (class 2): the four envs of a packed wavefront then mostly walk the same contact code paths */
/* longest first: the threshold of this step -- the override, or what the previous step's plan derived from its histogram */
constexpr int LPT_BINS = 128, LPT_BIN_SHIFT = 11;          /* bins of 2048 x 64 cycles = 0.06 ms; 128 of them: 7.6 ms */
__device__ __forceinline__ int lpt_threshold(const EnvParams& P)
{
    return P.lpt_thresh > 0 ? P.lpt_thresh : (P.lpt_permille > 0 ? P.lpt_state[P.lpt_parity] : 0);
}
__device__ __forceinline__ int plan_class(const EnvParams& P, const float* actions, int env, const PlanIn& in)
{
    if (contact_prone(P, actions, env, in)) return 0;
    if (P.nb == 0) return 1;
    if (P.nb > 1 && P.env_cycles && (P.lpt_thresh > 0 || P.lpt_permille > 0)) {
        const int thr = lpt_threshold(P);
        if (thr > 0) return P.env_cycles[2 * env] > thr ? 1 : 2;
    }
    float z = in.ee[2];
    float zn = fminf(fmaxf(z + in.a[2] * 0.01f, P.ee_lo[2]), P.ee_hi[2]);
    return fminf(z, zn) < P.ee_lo[2] + 0.012f ? 1 : 2;
}
__device__ __forceinline__ int plan_class(const EnvParams& P, const float* actions, int env)
{
    return plan_class(P, actions, env, plan_fetch(P, actions, env));
}
/* one word behind the per-workgroup counts of the two-pass plan: did the plan move the fingers-down class to list 0?
 * (then list 0 carries the long pole of the step and its wavefronts take issue priority, pmg_k_step_list) */
__device__ __forceinline__ int* plan_promoted(const EnvParams& P)
{
    return P.sched + 3 + 3 * (size_t)P.n_envs + 3 * (size_t)((P.n_envs + 1023) / 1024);
}
/* SIMPLE: reach under tip control -- the class is one compare on the tip target's height; its own instantiation keeps
 * the general classification (joint-control FK, object and chest tests: 5 000 instructions nobody executes for reach)
 * out of the kernel */
template <bool SIMPLE = false>
__device__ __forceinline__ void plan_all(const EnvParams& P, const float* actions)
{
    __shared__ int cnt0[PLAN_MAX_TILES], cnt1[PLAN_MAX_TILES], cnt2[PLAN_MAX_TILES];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = PLAN_THREADS / 64;
    const int chunks = (P.n_envs + PLAN_THREADS - 1) / PLAN_THREADS;
    const unsigned long long below = (1ull << lane) - 1ull;
    /* every env is classified ONCE (two bits per chunk kept in a register; the second pass used to classify again: eight
     * rounds of dependent global loads for 4096 envs, 16 us), and the loads of four chunks are in flight together */
    unsigned clsbits = 0;                                   /* chunks <= PLAN_SINGLE_MAX / PLAN_THREADS = 16 */
    static_assert(PLAN_SINGLE_MAX / PLAN_THREADS <= 16, "two class bits per chunk in one register");
    for (int c0 = 0; c0 < chunks && c0 < 16; c0 += 4) {     /* (a caller beyond PLAN_SINGLE_MAX: chunks 16.. are classified below) */
        PlanIn in[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int env = (c0 + k) * PLAN_THREADS + tid;
            in[k] = plan_fetch<SIMPLE>(P, actions, env < P.n_envs ? env : 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int env = (c0 + k) * PLAN_THREADS + tid;
            int cls;
            if (SIMPLE) {
                const float z = in[k].ee[2], zn = fminf(fmaxf(z + in[k].a[2] * 0.01f, P.ee_lo[2]), P.ee_hi[2]);
                cls = fminf(z, zn) < P.ee_lo[2] + 0.012f ? 0 : 1;        /* = plan_class for nb == 0, tip control */
            } else cls = plan_class(P, actions, env < P.n_envs ? env : 0, in[k]);   /* (lanes beyond the batch classify env 0: no read past the buffers) */
            cls = env < P.n_envs ? cls : 3;
            clsbits |= (unsigned)cls << (2 * (c0 + k));
        }
    }
    for (int c = 0; c < chunks; c++) {
        const int env = c * PLAN_THREADS + tid;
        const int cls = c < 16 ? (int)((clsbits >> (2 * c)) & 3u) : (env < P.n_envs ? plan_class(P, actions, env) : 3);
        unsigned long long m0 = wv::ballot(cls == 0), m1 = wv::ballot(cls == 1), m2 = wv::ballot(cls == 2);
        if (lane == 0 && c * waves + wave < PLAN_MAX_TILES) {
            cnt0[c * waves + wave] = __popcll(m0); cnt1[c * waves + wave] = __popcll(m1); cnt2[c * waves + wave] = __popcll(m2);
        }
    }
    __syncthreads();
    const int tiles = chunks * waves < PLAN_MAX_TILES ? chunks * waves : PLAN_MAX_TILES;
    /* exclusive prefix over the tiles, in place, one thread per class (eight counts per LDS round trip) -- every thread
     * used to add up the tiles before its own, 1024 threads x up to 64 tiles x 3 classes on ONE compute unit's LDS: most
     * of the kernel's 16 us */
    __shared__ int tot[3];
    if (lane == 0 && wave < 3) {
        int* cn = wave == 0 ? cnt0 : (wave == 1 ? cnt1 : cnt2);
        int run = 0;
        for (int t = 0; t < tiles; t += 8) {
            int v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = t + k < tiles ? cn[t + k] : 0;
#pragma unroll
            for (int k = 0; k < 8; k++) { if (t + k < tiles) cn[t + k] = run; run += v[k]; }
        }
        tot[wave] = run;
    }
    __syncthreads();
    const int n0all = tot[0], n1 = tot[1], n2all = tot[2];   /* class 2 starts behind all of class 1 in the second list */
    /* one free object, fingers down at the table (class 1: 8 more contacts = 24 more rows per sweep): such an env holds
     * its packed wavefront back for the whole step, alone on a wavefront it solves them in row space.  Worth a wavefront
     * each only while they are few (pick_and_place: +12 %; push / slide, where a fifth of the batch is down there at
     * any time: -24 %), so pick_and_place moves the whole class to the first list, behind class 0, and push / slide do not.
     * Rounds 2-4 decided it per step from batch-wide counts (under 1 / PMG_FD_DIV of the batch AND the step within 1.5
     * wavefronts per SIMD): the two kernels sum in different float32 orders, so an env's trajectory depended on the batch it
     * was in -- a run on 8 GPUs was not the run on 1 GPU.  Round 5: a rule of the task (EnvParams::fd_div) */
    const bool promote = P.nb == 1 && !P.joint_control &&
                         (P.fd_div < 0 || (P.fd_div > 0 && (long long)n1 * P.fd_div <= P.n_envs && n0all + n1 + ((n2all + 3) >> 2) <= P.wave_budget));
    for (int c = 0; c < chunks; c++) {
        int tile = c * waves + wave;
        int env = c * PLAN_THREADS + tid;
        int cls = (env < P.n_envs && tile < PLAN_MAX_TILES) ? (c < 16 ? (int)((clsbits >> (2 * c)) & 3u) : plan_class(P, actions, env)) : 3;
        unsigned long long m0 = wv::ballot(cls == 0), m1 = wv::ballot(cls == 1), m2 = wv::ballot(cls == 2);
        const int tt = tile < tiles ? tile : 0;
        const int b0 = cnt0[tt], b1 = cnt1[tt], b2 = cnt2[tt];
        if (cls == 0) P.sched[2 + b0 + __popcll(m0 & below)] = env;
        else if (cls == 1 && promote) P.sched[2 + n0all + b1 + __popcll(m1 & below)] = env;
        else if (cls == 1) P.sched[2 + P.n_envs + b1 + __popcll(m1 & below)] = env;
        else if (cls == 2) P.sched[2 + P.n_envs + (promote ? 0 : n1) + b2 + __popcll(m2 & below)] = env;
    }
    if (tid == 0) {
        const int n0 = n0all, n2 = n2all;
        P.sched[0] = promote ? n0 + n1 : n0;
        P.sched[1] = promote ? n2 : n1 + n2;
        P.sched[2 + 2 * P.n_envs] = 0; /* redo list of the fast paths starts empty */
        *plan_promoted(P) = promote ? 1 : 0;
    }
}
/* Batches beyond one plan workgroup (65 536 envs): the same stable three-way partition in two passes over
 * ceil(N / 1024) workgroups.  Pass 1 leaves every workgroup's class counts behind the schedule; pass 2 re-derives the
 * classes (a few loads and compares per env), turns the counts of the workgroups before it into its bases and the
 * grand totals into the promotion decision -- the SAME decision in every workgroup, so the lists come out exactly as
 * the single-workgroup plan would write them -- and scatters. */
__device__ __forceinline__ int* plan_wg_counts(const EnvParams& P) { return P.sched + 3 + 3 * (size_t)P.n_envs; }
__device__ __forceinline__ void plan_count(const EnvParams& P, const float* actions)
{
    __shared__ int acc[3];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    if (tid < 3) acc[tid] = 0;
    __syncthreads();
    const int env = (int)blockIdx.x * PLAN_THREADS + tid;
    const int cls = env < P.n_envs ? plan_class(P, actions, env) : -1;
    const unsigned long long m0 = wv::ballot(cls == 0), m1 = wv::ballot(cls == 1), m2 = wv::ballot(cls == 2);
    if (lane == 0) { atomicAdd(&acc[0], __popcll(m0)); atomicAdd(&acc[1], __popcll(m1)); atomicAdd(&acc[2], __popcll(m2)); }
    const bool lpt = P.nb > 1 && P.lpt_permille > 0 && P.env_cycles && P.lpt_thresh <= 0;   /* (uniform) */
    __shared__ int hist[LPT_BINS];
    if (lpt) {
        if (tid < LPT_BINS) hist[tid] = 0;
        __syncthreads();
        if (cls > 0) {                                       /* a fast-path env: its wavefront time of the last step */
            const int b = P.env_cycles[2 * env] >> LPT_BIN_SHIFT;
            atomicAdd(&hist[b < LPT_BINS ? b : LPT_BINS - 1], 1);
        }
    }
    __syncthreads();
    if (tid < 3) plan_wg_counts(P)[3 * (int)blockIdx.x + tid] = acc[tid];
    if (lpt && tid < LPT_BINS && hist[tid]) atomicAdd(&P.lpt_state[2 + tid], hist[tid]);
}
__device__ __forceinline__ void plan_scatter(const EnvParams& P, const float* actions)
{
    __shared__ int tot[3], base[3], wcnt[3][PLAN_THREADS / 64];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = PLAN_THREADS / 64;
    const int wg = (int)blockIdx.x, nwg = (P.n_envs + PLAN_THREADS - 1) / PLAN_THREADS;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (tid < 3) { tot[tid] = 0; base[tid] = 0; }
    __syncthreads();
    {
        const int* cnt = plan_wg_counts(P);
        int t0 = 0, t1 = 0, t2 = 0, b0 = 0, b1 = 0, b2 = 0;
        for (int w = tid; w < nwg; w += PLAN_THREADS) {
            const int c0 = cnt[3 * w], c1 = cnt[3 * w + 1], c2 = cnt[3 * w + 2];
            t0 += c0; t1 += c1; t2 += c2;
            if (w < wg) { b0 += c0; b1 += c1; b2 += c2; }
        }
        if (t0 | t1 | t2) { atomicAdd(&tot[0], t0); atomicAdd(&tot[1], t1); atomicAdd(&tot[2], t2); }
        if (b0 | b1 | b2) { atomicAdd(&base[0], b0); atomicAdd(&base[1], b1); atomicAdd(&base[2], b2); }
    }
    const int env = wg * PLAN_THREADS + tid;
    const int cls = env < P.n_envs ? plan_class(P, actions, env) : -1;
    const unsigned long long m0 = wv::ballot(cls == 0), m1 = wv::ballot(cls == 1), m2 = wv::ballot(cls == 2);
    if (lane == 0) { wcnt[0][wave] = __popcll(m0); wcnt[1][wave] = __popcll(m1); wcnt[2][wave] = __popcll(m2); }
    __syncthreads();
    const int n0all = tot[0], n1 = tot[1], n2all = tot[2];
    const bool promote = P.nb == 1 && !P.joint_control &&
                         (P.fd_div < 0 || (P.fd_div > 0 && (long long)n1 * P.fd_div <= P.n_envs && n0all + n1 + ((n2all + 3) >> 2) <= P.wave_budget));
    int b0 = base[0], b1 = base[1], b2 = base[2];
    for (int w = 0; w < wave && w < waves; w++) { b0 += wcnt[0][w]; b1 += wcnt[1][w]; b2 += wcnt[2][w]; }
    if (cls == 0) P.sched[2 + b0 + __popcll(m0 & below)] = env;
    else if (cls == 1 && promote) P.sched[2 + n0all + b1 + __popcll(m1 & below)] = env;
    else if (cls == 1) P.sched[2 + P.n_envs + b1 + __popcll(m1 & below)] = env;
    else if (cls == 2) P.sched[2 + P.n_envs + (promote ? 0 : n1) + b2 + __popcll(m2 & below)] = env;
    if (wg == 0 && tid == 0) {
        P.sched[0] = promote ? n0all + n1 : n0all;
        P.sched[1] = promote ? n2all : n1 + n2all;
        P.sched[2 + 2 * P.n_envs] = 0; /* redo list of the fast paths starts empty */
        *plan_promoted(P) = promote ? 1 : 0;
    }
    if (wg == 0 && P.nb > 1 && P.lpt_permille > 0 && P.env_cycles && P.lpt_thresh <= 0) {
        /* the NEXT step's threshold (the other parity's word: workgroups of this launch still read this step's): walk the
         * histogram of the fast-path envs' cycles from the top until the slowest lpt_permille / 1000 of them are counted */
        __shared__ int hs[LPT_BINS];
        if (tid < LPT_BINS) { hs[tid] = P.lpt_state[2 + tid]; P.lpt_state[2 + tid] = 0; }
        __syncthreads();
        if (tid == 0) {
            long long total = 0;
            for (int b = 0; b < LPT_BINS; b++) total += hs[b];
            long long acc2 = 0;
            int b = LPT_BINS - 1;
            for (; b > 0; b--) { acc2 += hs[b]; if (acc2 * 1000 >= total * P.lpt_permille) break; }
            P.lpt_state[P.lpt_parity ^ 1] = (hs[0] == total) ? 0 : (b << LPT_BIN_SHIFT);   /* (nothing measured yet: off) */
        }
    }
}
__device__ __forceinline__ int scheduled_env(const EnvParams& P, int block)
{
    int n0 = P.sched[0];
    return block < n0 ? P.sched[2 + block] : P.sched[2 + P.n_envs + (block - n0)];
}

/* ------------------------------------------------------------------ */
/* MT19937 exactly as numpy's RandomState (state in HBM, used by lane 0 only) */
__device__ __forceinline__ unsigned mt_next(unsigned* mt)
{
    unsigned idx = mt[624];
    if (idx >= 624u) {
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            unsigned y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < 623; kk++) {
            unsigned y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        unsigned y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        idx = 0;
    }
    unsigned y = mt[idx];
    mt[624] = idx + 1;
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ double mt_uniform(unsigned* mt, double lo, double hi)
{
    unsigned a = mt_next(mt) >> 5, b = mt_next(mt) >> 6;
    double u = (a * 67108864.0 + b) / 9007199254740992.0;
    return lo + (hi - lo) * u;
}
__device__ __forceinline__ unsigned mt_interval(unsigned* mt, unsigned mx)
{
    if (mx == 0) return 0;
    unsigned mask = mx;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    unsigned v;
    while ((v = (mt_next(mt) & mask)) > mx) {}
    return v;
}

/* task reset by lane 0: kuka_single_step_base_env.py:76-148, kuka_multi_step_base_env.py:221-250,
 * kuka_multi_step_envs.py:34-87 */
__device__ __forceinline__ void task_reset_lane0(const EnvParams& P, int env)
{
    unsigned* mt = P.rng + (size_t)env * 625;
    float* g = P.goal + (size_t)env * GOAL_DIM;
    float* cold = P.cold + (size_t)env * COLD_DIM;
    float* blk = P.blocks + (size_t)env * BLOCK_DIM * P.nb;
    if (!P.multi) {
        double center[3] = {P.tip_init[0], P.tip_init[1], P.tip_init[2]};
        if (P.has_obj) {
            double ox = P.tip_init[0], oy = P.tip_init[1];
            for (;;) {
                double dx = ox - P.tip_init[0], dy = oy - P.tip_init[1];
                if (!(sqrt(dx * dx + dy * dy) < 0.1)) break;
                ox = mt_uniform(mt, P.obj_lo[0], P.obj_hi[0]);
                oy = mt_uniform(mt, P.obj_lo[1], P.obj_hi[1]);
            }
            blk[0] = (float)ox; blk[1] = (float)oy; blk[2] = (float)P.obj_z;
            blk[3] = 0.f; blk[4] = 0.f; blk[5] = 0.f; blk[6] = 1.f;
            for (int a = 7; a < 13; a++) blk[a] = 0.f;
            center[0] = ox; center[1] = oy; center[2] = P.obj_z;
        }
        double gg[3];
        for (;;) {
            for (int a = 0; a < 3; a++) gg[a] = mt_uniform(mt, P.tgt_lo[a], P.tgt_hi[a]);
            double dx = gg[0] - center[0], dy = gg[1] - center[1], dz = gg[2] - center[2];
            if (sqrt(dx * dx + dy * dy + dz * dz) > 0.1) break;
        }
        if (!P.in_air) gg[2] = P.obj_z;
        else if (P.grasping) {
            if (mt_uniform(mt, 0.0, 1.0) >= 0.5) gg[2] = P.obj_z;
        }
        for (int a = 0; a < 3; a++) g[a] = (float)gg[a];
    } else {
        double bp[5][2];
        for (int b = 0; b < P.nb; b++) {
            for (;;) {
                double x = mt_uniform(mt, P.obj_lo[0], P.obj_hi[0]);
                double y = mt_uniform(mt, P.obj_lo[1], P.obj_hi[1]);
                bool ok = true;
                for (int cc = 0; cc < b; cc++) {
                    double dx = x - bp[cc][0], dy = y - bp[cc][1];
                    if (!(sqrt(dx * dx + dy * dy) > 0.06)) ok = false;
                }
                double dx = x - P.tip_init[0], dy = y - P.tip_init[1];
                if (!(sqrt(dx * dx + dy * dy) > 0.06)) ok = false;
                if (ok) { bp[b][0] = x; bp[b][1] = y; break; }
            }
        }
        for (int b = 0; b < P.nb; b++) {
            float* o = blk + BLOCK_DIM * b;
            o[0] = (float)bp[b][0]; o[1] = (float)bp[b][1]; o[2] = 0.175f;
            o[3] = 0.f; o[4] = 0.f; o[5] = 0.f; o[6] = 1.f;
            for (int a = 7; a < 13; a++) o[a] = 0.f;
        }
        int order[5] = {0, 1, 2, 3, 4};
        /* sub_goal_ind = -1 after reset (kuka_multi_step_base_env.py:248-249): the last sub-goal */
        int level = (P.grip_goal && P.decomposition) ? 2 * P.nb - 1 : P.nb - 1, moved = (1 << P.nb) - 1;
        if (P.chest >= 0) {
            /* chest_robot.robot_specific_reset (kuka_multi_step_base_env.py:242-243): door closed, at rest, motor off; the
             * goal is the chest itself, nothing is drawn.  sub_goal_ind = -1 = the last of num_steps
             * (kuka_multi_step_envs.py:238-242, 388-392) */
            for (int a = 0; a < GOAL_DIM; a++) g[a] = 0.f;
            level = (P.grip_goal ? P.nb * (P.grasping ? 3 : 2) : P.nb);
            moved = (1 << P.nb) - 1;
        } else if (P.task == PMG_TASK_BLOCK_STACK) {
            if (P.random_order)
                for (int i = P.nb - 1; i >= 1; i--) {
                    unsigned j = mt_interval(mt, (unsigned)i);
                    int t = order[i]; order[i] = order[j]; order[j] = t;
                }
            double bx, by;
            for (;;) {
                bx = mt_uniform(mt, P.tgt_lo[0], P.tgt_hi[0]);
                by = mt_uniform(mt, P.tgt_lo[1], P.tgt_hi[1]);
                bool ok = true;
                for (int cc = 0; cc < P.nb; cc++) {
                    double dx = bx - bp[cc][0], dy = by - bp[cc][1];
                    if (!(sqrt(dx * dx + dy * dy) > 0.08)) ok = false;
                }
                if (ok) break;
            }
            cold[13] = (float)bx; cold[14] = (float)by; cold[15] = 0.175f;
            for (int s = 0; s < P.nb; s++) {
                int b = order[s];
                g[3 * b] = (float)bx; g[3 * b + 1] = (float)by; g[3 * b + 2] = 0.175f + 0.03f * (float)s;
            }
        } else { /* block_rearrange: one table target per block, clear of targets and blocks (kuka_multi_step_envs.py:174-190) */
            double tp[5][2];
            for (int t = 0; t < P.nb; t++)
                for (;;) {
                    double x = mt_uniform(mt, P.tgt_lo[0], P.tgt_hi[0]);
                    double y = mt_uniform(mt, P.tgt_lo[1], P.tgt_hi[1]);
                    bool ok = true;
                    for (int cc = 0; cc < t; cc++) {
                        double dx = x - tp[cc][0], dy = y - tp[cc][1];
                        if (!(sqrt(dx * dx + dy * dy) > 0.06)) ok = false;
                    }
                    for (int cc = 0; cc < P.nb; cc++) {
                        double dx = x - bp[cc][0], dy = y - bp[cc][1];
                        if (!(sqrt(dx * dx + dy * dy) > 0.06)) ok = false;
                    }
                    if (ok) { tp[t][0] = x; tp[t][1] = y; break; }
                }
            for (int t = 0; t < P.nb; t++) { g[3 * t] = (float)tp[t][0]; g[3 * t + 1] = (float)tp[t][1]; g[3 * t + 2] = 0.175f; }
        }
        if (P.curriculum) {
            /* level = np_random.choice(num_curriculum, p=curriculum_prob): normalised cdf, one double draw,
             * searchsorted(side='right')  (kuka_multi_step_envs.py:128, 202) */
            /* num_curriculum levels: num_block, or num_block + 1 for the chest tasks (level = how many blocks go in);
             * row layout prob[NC] | generated[NC] | goal_step with NC = 5 / 6 */
            const int ncur = P.chest >= 0 ? P.nb + 1 : P.nb, NC = P.chest >= 0 ? 6 : 5;
            float* cs = P.curr + (size_t)env * CURR_DIM;
            double cdf[6], acc = 0.0;
            for (int i = 0; i < ncur; i++) { acc += (double)cs[i]; cdf[i] = acc; }
            double u = mt_uniform(mt, 0.0, 1.0);
            level = 0;
            while (level < ncur && cdf[level] / acc <= u) level++;
            if (level > ncur - 1) level = ncur - 1;
            cs[2 * NC] = (float)(level * 25 + 50);
            if (P.task == PMG_TASK_BLOCK_REARRANGE || P.chest >= 0) {
                /* choice(arange(nb), size=level+1 (rearrange) / level (chest), replace=False) = permutation(nb)[:size] */
                int perm[5] = {0, 1, 2, 3, 4};
                for (int i = P.nb - 1; i >= 1; i--) {
                    unsigned j = mt_interval(mt, (unsigned)i);
                    int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
                }
                moved = 0;
                const int take = P.chest >= 0 ? level : level + 1;
                for (int i = 0; i < take; i++) moved |= 1 << perm[i];
            }
            if (P.curriculum_update) { /* _update_curriculum_prob: kuka_multi_step_base_env.py:350-379 */
                cs[NC + level] += 1.f;
                bool fin[6], half[6];
                for (int i = 0; i < ncur; i++) {
                    fin[i] = (double)cs[NC + i] >= P.goals_per_curriculum;
                    half[i] = (double)cs[NC + i] >= P.goals_per_curriculum / 2;
                    if (fin[i]) cs[i] = 0.f;
                }
                if (half[0] && !fin[0]) { cs[0] = 0.5f; cs[1] = 0.5f; }
                for (int i = 1; i < ncur - 1; i++)
                    if (fin[i - 1] && !fin[i]) {
                        if (half[i]) { cs[i] = 0.5f; cs[i + 1] = 0.5f; }
                        else cs[i] = 1.f;
                    }
                if (fin[ncur - 2]) cs[ncur - 1] = 1.f;
            }
        }
        for (int b = 0; b < 5; b++) cold[8 + b] = (float)order[b];
        cold[7] = (float)level;
        g[15] = (float)moved;
    }
}

/* env.reset(): kuka.py:120-165 + the task reset above + _get_obs */
/* done_only (pmg_reset_done_device): the envs whose episode has ended by TimeLimit reset themselves, every other workgroup
 * leaves on its first load; the packed row keeps the finished step's reward | goal_achieved | done */
__device__ __forceinline__ void reset_env(const EnvParams& P, const unsigned char* mask, int done_only = 0)
{
    __shared__ LaneTabStore lcs;
    int env = (int)blockIdx.x, l = wv::lane();
    if (env >= P.n_envs) return;
    float* hot = P.hot + (size_t)env * HOT_DIM;
    if (done_only && (int)hot[29] < P.max_steps) return;
    LaneConst c;
    load_lane_const(lcs, c);
    float* cold = P.cold + (size_t)env * COLD_DIM;
    int ll = l < NJ ? l : 0;
    bool doit = done_only || mask == nullptr || mask[env] != 0;
    float q, qd;
    int elapsed;
    if (doit) {
        q = l < 7 ? cold[l] : (l < NJ ? FINGER_LIMIT : 0.f); /* kuka.py:158,161 */
        qd = 0.f;
        float tgt[3] = {(float)P.tip_init[0], (float)P.tip_init[1], (float)P.tip_init[2]};
        /* the reference's constructor resets the robot once on its own (base_env.py:42) before its first env.reset()
         * (:84): a world's very first reset runs the reset IK twice, the second from the first's solution (kuka.py:159-160;
         * tests/golden/ref_*.json) */
        if (hot[31] == 0.f) {
            float q1 = ik_solve(c, q, tgt);
            if (l < 7) q = q1;
        }
        float qik = ik_solve(c, q, tgt);                       /* kuka.py:159 */
        if (l < 7) { q = qik; cold[l] = qik; }                /* kuka.py:160 */
        Kin k;
        fk(c, q, k);
        float tip[3], Rt[9];
        tip_frame(k, tip, Rt);
        if (l < NJ) { hot[l] = q; hot[9 + l] = 0.f; }
        if (l < 3) hot[18 + l] = l == 0 ? tip[0] : (l == 1 ? tip[1] : tip[2]); /* kuka.py:163 */
        if (l < 7) hot[21 + l] = q;                            /* kuka.py:165 */
        if (l == 0) {
            hot[28] = FINGER_LIMIT; hot[29] = 0.f; hot[30] = 0.f; hot[31] = hot[31] + 1.f;
            task_reset_lane0(P, env);
        }
        elapsed = 0;
        wv::lds_sync();
    } else {
        q = l < NJ ? hot[ll] : 0.f;
        qd = l < NJ ? hot[9 + ll] : 0.f;
        elapsed = (int)hot[29];
    }
    write_outputs(P, env, c, q, qd, elapsed, (doit && !done_only) ? TAIL_ZERO : TAIL_KEEP);
}

}  // namespace pmg
#include "pmg_packed.h"
#endif
