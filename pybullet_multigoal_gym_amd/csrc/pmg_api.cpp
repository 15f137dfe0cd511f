/*
 * pmg_api.cpp -- host side of the C ABI declared in include/pmg.h.
 *
 * Owns the device arrays of N environments on one MI355X, seeds them like
 * gym.utils.seeding.np_random (sha512 -> MT19937 init_by_array), launches
 * the step / reset kernels on its own HIP stream, times the step kernel with
 * HIP events, and exchanges the packed observation shard with RCCL.
 * There is no CPU fallback: without a HIP device pmg_create fails.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pmg.h"
#include "../../include/pmg_sha512_const.h"
#include "pmg_launch.h"

namespace {

char g_create_err[512] = "";

constexpr int EVENT_POOL = 2048;

struct Seeder { /* gym 0.17.3 seeding.np_random + numpy RandomState.seed(int list) */
    static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    static void sha512(const unsigned char* msg, size_t len, unsigned char out[64])
    {
        static const uint64_t K[80] = PMG_SHA512_K;
        uint64_t h[8] = PMG_SHA512_H0;
        size_t total = ((len + 17 + 127) / 128) * 128;
        std::vector<unsigned char> buf(total, 0);
        memcpy(buf.data(), msg, len);
        buf[len] = 0x80;
        uint64_t bits = (uint64_t)len * 8;
        for (int i = 0; i < 8; i++) buf[total - 1 - i] = (unsigned char)(bits >> (8 * i));
        for (size_t off = 0; off < total; off += 128) {
            uint64_t w[80];
            for (int i = 0; i < 16; i++) {
                uint64_t v = 0;
                for (int b = 0; b < 8; b++) v = (v << 8) | buf[off + 8 * i + b];
                w[i] = v;
            }
            for (int i = 16; i < 80; i++) {
                uint64_t s0 = rotr(w[i - 15], 1) ^ rotr(w[i - 15], 8) ^ (w[i - 15] >> 7);
                uint64_t s1 = rotr(w[i - 2], 19) ^ rotr(w[i - 2], 61) ^ (w[i - 2] >> 6);
                w[i] = w[i - 16] + s0 + w[i - 7] + s1;
            }
            uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
            for (int i = 0; i < 80; i++) {
                uint64_t t1 = hh + (rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
                uint64_t t2 = (rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
                hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
            }
            h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
        }
        for (int i = 0; i < 8; i++)
            for (int b = 0; b < 8; b++) out[8 * i + b] = (unsigned char)(h[i] >> (56 - 8 * b));
    }
    /* fills mt[0..623] and mt[624] = 624 (index) */
    static void seed(uint32_t* mt, uint64_t seed)
    {
        char txt[32];
        int n = snprintf(txt, sizeof(txt), "%llu", (unsigned long long)seed);
        unsigned char dig[64];
        sha512((const unsigned char*)txt, (size_t)n, dig);
        uint32_t key[2];
        for (int w = 0; w < 2; w++)
            key[w] = (uint32_t)dig[4 * w] | ((uint32_t)dig[4 * w + 1] << 8) | ((uint32_t)dig[4 * w + 2] << 16) | ((uint32_t)dig[4 * w + 3] << 24);
        int klen = key[1] ? 2 : 1;
        mt[0] = 19650218u;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        int i = 1, j = 0;
        for (int k = 624; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            i++; j++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
            if (j >= klen) j = 0;
        }
        for (int k = 623; k; k--) {
            mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
            i++;
            if (i >= 624) { mt[0] = mt[623]; i = 1; }
        }
        mt[0] = 0x80000000u;
        mt[624] = 624u;
    }
};

}  // namespace

struct pmg_env {
    pmg_config cfg;
    pmg_dims dims;
    pmg::EnvParams P;
    int nb;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;       /* second queue: the full-store list of the multi-block tasks runs beside the main one */
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float* d_actions = nullptr;       /* staging for host-buffer pmg_step */
    unsigned char* d_mask = nullptr;
    float* h_packed = nullptr;        /* pinned */
    float* h_actions = nullptr;       /* pinned */
    float* d_rw_ag = nullptr; float* d_rw_dg = nullptr; float* d_rw_r = nullptr; unsigned char* d_rw_ok = nullptr;
    long long rw_cap = 0;
    hipEvent_t ev_a[EVENT_POOL], ev_b[EVENT_POOL];
    hipEvent_t cv_a[EVENT_POOL], cv_b[EVENT_POOL];   /* the same around every all-gather */
    int cv_n = 0;
    double cv_ms = 0.0, cv_max = 0.0;
    long long cv_launches = 0;
    int ev_n = 0;
    int ev_every = 1;                 /* events around every ev_every-th batched step (pmg_timing_every) */
    long long step_count = 0;
    long long plans = 0;              /* batched steps planned so far (parity of the longest-first threshold words) */
    double ev_ms = 0.0, ev_min = 0.0, ev_max = 0.0;
    long long ev_launches = 0;
    bool ever_reset = false;
    int packed = 1;                   /* reach: contact-free envs four per wavefront (PMG_PACKED=0 switches it off) */
    int two_wave = 1;                 /* reach: the two-wavefront kernel for steps with contact-prone envs (PMG_REACH_TWO_WAVES=0: never) */
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    /* overlapped all-gather (pmg_comm_overlap): the packed rows are double-buffered -- step t writes out2[t & 1], its all-gather
     * reads that buffer on the communication stream while step t + 1 computes and writes the OTHER one; step t + 2, which
     * writes out2[t & 1] again, is the one that waits for gather t */
    bool overlap = false;
    float* out2[2] = {nullptr, nullptr};
    int out_phase = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_rows[2] = {nullptr, nullptr}, ev_gdone[2] = {nullptr, nullptr};
    bool gpending[2] = {false, false};
    int last_gather = -1;
    char err[512] = "";
};

#ifndef PMG_LPT_PERMILLE_DEFAULT
#define PMG_LPT_PERMILLE_DEFAULT 400   /* the slowest 40 % of the fast-path list lead it (profiles/r06_lpt_quantile_sweep.txt) */
#endif

namespace {

int fail(pmg_env* e, int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e ? e->err : g_create_err, 512, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(e, call)                                                                             \
    do {                                                                                             \
        hipError_t rc_ = (call);                                                                     \
        if (rc_ != hipSuccess) return fail(e, PMG_E_DEVICE, "%s -> %s", #call, hipGetErrorString(rc_)); \
    } while (0)

int fill_dims(const pmg_config* c, pmg_dims* d, int* nb_out)
{
    memset(d, 0, sizeof(*d));
    int jo = c->joint_control ? 7 : 0;
    d->num_envs = c->num_envs;
    switch (c->task) {
    case PMG_TASK_REACH: d->action_dim = jo ? 7 : 3; d->observation_dim = 3 + jo; d->policy_state_dim = 3 + jo; d->goal_dim = 3; break;
    case PMG_TASK_PUSH: case PMG_TASK_SLIDE: d->action_dim = jo ? 7 : 3; d->observation_dim = 20 + jo; d->policy_state_dim = 7 + jo; d->goal_dim = 3; break;
    case PMG_TASK_PICK_AND_PLACE: d->action_dim = jo ? 8 : 4; d->observation_dim = 20 + jo; d->policy_state_dim = 7 + jo; d->goal_dim = 3; break;
    case PMG_TASK_BLOCK_STACK:
    case PMG_TASK_BLOCK_REARRANGE:
        if (c->num_block < 1 || c->num_block > 5) return -1;
        if (c->use_curriculum && (c->num_block < 2 || c->task_decomposition)) return -1; /* kuka_multi_step_base_env.py:123,131 */
        if (c->task_decomposition && c->task != PMG_TASK_BLOCK_STACK) return -1;          /* kuka_multi_step_envs.py:159 */
        d->action_dim = (jo ? 7 : 3) + (c->task == PMG_TASK_BLOCK_STACK ? 1 : 0); d->observation_dim = 8 + 16 * c->num_block + jo;
        d->policy_state_dim = 4 + 3 * c->num_block + jo; d->goal_dim = 3 * c->num_block;
        if (c->grip_informed_goal) {
            if (c->task != PMG_TASK_BLOCK_STACK) return -1; /* kuka_multi_step_envs.py:158 */
            d->goal_dim += 4;                               /* tip xyz + finger width, kuka_multi_step_base_env.py:300-304 */
        }
        break;
    case PMG_TASK_CHEST_PUSH:
    case PMG_TASK_CHEST_PICK_AND_PLACE: {
        /* kuka_multi_step_base_env.py:283-304: the multi-block layout + door joint pos / vel + 3 key points x (xyz, vel);
         * goals lead with the door state; num_curriculum = num_block + 1 (kuka_multi_step_envs.py:253,402) */
        if (c->num_block < 1 || c->num_block > 5 || (c->use_curriculum && c->task_decomposition)) return -1;
        const int gr = c->task == PMG_TASK_CHEST_PICK_AND_PLACE ? 1 : 0;
        d->action_dim = (jo ? 7 : 3) + gr; d->observation_dim = 8 + 16 * c->num_block + jo + 20;
        d->policy_state_dim = 4 + 3 * c->num_block + jo + 19; d->goal_dim = 1 + 3 * c->num_block;
        if (c->grip_informed_goal) d->goal_dim += gr ? 4 : 3;
        break;
    }
    default: return -1;
    }
    bool multi = c->task == PMG_TASK_BLOCK_STACK || c->task == PMG_TASK_BLOCK_REARRANGE || c->task == PMG_TASK_CHEST_PUSH ||
                 c->task == PMG_TASK_CHEST_PICK_AND_PLACE;
    if (!multi && (c->use_curriculum || c->task_decomposition || c->grip_informed_goal)) return -1;
    int nb = c->task == PMG_TASK_REACH ? 0 : (multi ? c->num_block : 1);
    *nb_out = nb;
    d->state_dim = 64 + 13 * nb + (c->use_curriculum ? pmg::CURR_DIM : 0);
    d->packed_dim = d->observation_dim + d->policy_state_dim + 2 * d->goal_dim + 3;
    return 0;
}

/* workspace constants: kuka.py:35-51, kuka_single_step_base_env.py:48-56 */
void fill_params(pmg_env* e)
{
    pmg::EnvParams& P = e->P;
    const pmg_config& c = e->cfg;
    int t = c.task;
    P.n_envs = c.num_envs; P.task = t; P.nb = e->nb;
    P.chest = t == PMG_TASK_CHEST_PUSH ? 0 : (t == PMG_TASK_CHEST_PICK_AND_PLACE ? 1 : -1);
    P.grasping = (t == PMG_TASK_PICK_AND_PLACE || t == PMG_TASK_BLOCK_STACK || t == PMG_TASK_CHEST_PICK_AND_PLACE);
    P.has_obj = (t != PMG_TASK_REACH);
    P.in_air = (t == PMG_TASK_REACH || t == PMG_TASK_PICK_AND_PLACE || t == PMG_TASK_BLOCK_STACK || t == PMG_TASK_CHEST_PICK_AND_PLACE);
    P.joint_control = c.joint_control; P.binary_reward = c.binary_reward; P.max_steps = c.max_episode_steps;
    P.random_order = c.random_order;
    P.multi = (t == PMG_TASK_BLOCK_STACK || t == PMG_TASK_BLOCK_REARRANGE || P.chest >= 0);
    P.curriculum = c.use_curriculum; P.curriculum_update = 0;
    P.decomposition = c.task_decomposition; P.grip_goal = c.grip_informed_goal;
    {
        double total = c.num_goals_to_generate > 0 ? (double)c.num_goals_to_generate : 1e6;
        P.goals_per_curriculum = e->nb > 0 ? floor(total / (P.chest >= 0 ? e->nb + 1 : e->nb)) : total; /* kuka_multi_step_base_env.py:139 */
    }
    P.adim = e->dims.action_dim; P.odim = e->dims.observation_dim; P.pdim = e->dims.policy_state_dim;
    P.gdim = e->dims.goal_dim; P.packed = e->dims.packed_dim;
    P.thr = c.distance_threshold;
    bool on_table = (t == PMG_TASK_PUSH || t == PMG_TASK_SLIDE || t == PMG_TASK_BLOCK_REARRANGE || t == PMG_TASK_CHEST_PUSH); /* kuka_multi_step_envs.py:169,399 */
    double range = 0.15, trange = 0.15;
    if (t == PMG_TASK_SLIDE) { range = 0.1; trange = 0.2; } /* kuka_single_step_base_env.py:66-69 */
    if (P.chest >= 0) range = 0.1;                          /* kuka_multi_step_envs.py:251,400 */
    P.tip_init[0] = -0.52; P.tip_init[1] = 0.0; P.tip_init[2] = 0.25;
    if (on_table) P.tip_init[2] = 0.175 + 0.001;
    const double hi[3] = {-0.37, 0.20, 0.55}, lo[3] = {-0.67, -0.20, 0.175};
    for (int a = 0; a < 3; a++) {
        P.ee_hi[a] = (float)hi[a]; P.ee_lo[a] = (float)lo[a];
        P.obj_lo[a] = P.tip_init[a] - range; P.obj_hi[a] = P.tip_init[a] + range;
        P.tgt_lo[a] = P.tip_init[a] - trange; P.tgt_hi[a] = P.tip_init[a] + trange;
    }
    P.obj_lo[0] += 0.03; P.obj_hi[0] -= 0.03;
    P.tgt_lo[0] += 0.03; P.tgt_hi[0] -= 0.03;
    P.tgt_lo[2] = lo[2];
    if (P.chest >= 0) { /* kuka_multi_step_base_env.py:102-105 */
        P.obj_lo[0] += 0.05; P.obj_hi[0] += 0.05;
        P.obj_lo[1] -= 0.05; P.obj_hi[1] += 0.05;
    }
    P.obj_z = 0.175;
    const float th[3] = PMG_TABLE_HALF;
    P.table_c[0] = -0.52f; P.table_c[1] = 0.f; P.table_c[2] = 0.08f;
    for (int a = 0; a < 3; a++) P.table_h[a] = th[a];
    P.table_mu = (float)PMG_TABLE_FRICTION;
    if (t == PMG_TASK_SLIDE) { /* long table (table_long.urdf), kuka_single_step_base_env.py:53-56; the puck itself is pmg::ObjT<true> */
        const float lt[3] = PMG_LONG_TABLE_HALF;
        P.tgt_lo[0] -= 0.4; P.tgt_hi[0] -= 0.4;
        P.table_c[0] = -0.70f;
        for (int a = 0; a < 3; a++) P.table_h[a] = lt[a];
        P.table_mu = (float)PMG_LONG_TABLE_FRICTION;
        P.obj_z = 0.170;
    }
}

int upload_seeds(pmg_env* e)
{
    int N = e->cfg.num_envs;
    std::vector<uint32_t> mt((size_t)N * 625);
    for (int i = 0; i < N; i++)
        Seeder::seed(mt.data() + (size_t)i * 625, e->cfg.seed_base + e->cfg.seed_stride * (uint64_t)(i + e->cfg.env_index_offset));
    HIP_TRY(e, hipMemcpyAsync(e->P.rng, mt.data(), mt.size() * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return PMG_OK;
}

void unpack(const pmg_env* e, const float* packed, float* obs, float* pol, float* ag, float* dg, float* reward,
            uint8_t* ga, uint8_t* done)
{
    const pmg_dims& d = e->dims;
    int N = d.num_envs, S = d.packed_dim;
    for (int i = 0; i < N; i++) {
        const float* r = packed + (size_t)i * S;
        if (obs) memcpy(obs + (size_t)i * d.observation_dim, r, sizeof(float) * d.observation_dim);
        r += d.observation_dim;
        if (pol) memcpy(pol + (size_t)i * d.policy_state_dim, r, sizeof(float) * d.policy_state_dim);
        r += d.policy_state_dim;
        if (ag) memcpy(ag + (size_t)i * d.goal_dim, r, sizeof(float) * d.goal_dim);
        r += d.goal_dim;
        if (dg) memcpy(dg + (size_t)i * d.goal_dim, r, sizeof(float) * d.goal_dim);
        r += d.goal_dim;
        if (reward) reward[i] = r[0];
        if (ga) ga[i] = r[1] != 0.f;
        if (done) done[i] = r[2] != 0.f;
    }
}

void drain_events(pmg_env* e)
{
    for (int i = 0; i < e->ev_n; i++) {
        float ms = 0.f;
        (void)hipEventSynchronize(e->ev_b[i]);
        if (hipEventElapsedTime(&ms, e->ev_a[i], e->ev_b[i]) == hipSuccess) {
            if (e->ev_launches == 0 || ms < e->ev_min) e->ev_min = ms;
            if (e->ev_launches == 0 || ms > e->ev_max) e->ev_max = ms;
            e->ev_ms += ms; e->ev_launches++;
        }
    }
    e->ev_n = 0;
}

void drain_comm_events(pmg_env* e)
{
    for (int i = 0; i < e->cv_n; i++) {
        float ms = 0.f;
        (void)hipEventSynchronize(e->cv_b[i]);
        if (hipEventElapsedTime(&ms, e->cv_a[i], e->cv_b[i]) == hipSuccess) {
            if (ms > e->cv_max) e->cv_max = ms;
            e->cv_ms += ms; e->cv_launches++;
        }
    }
    e->cv_n = 0;
}

}  // namespace

extern "C" {

const char* pmg_last_error(const pmg_env* e) { return e ? e->err : g_create_err; }

int pmg_create(const pmg_config* cfg, pmg_env** out)
{
    if (!cfg || !out) return fail(nullptr, PMG_E_INVALID, "pmg_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(pmg_config)) return fail(nullptr, PMG_E_INVALID, "pmg_create: struct_size %d != %zu", cfg->struct_size, sizeof(pmg_config));
    if (cfg->num_envs < 1) return fail(nullptr, PMG_E_INVALID, "pmg_create: num_envs must be >= 1");
    if (cfg->max_episode_steps < 1) return fail(nullptr, PMG_E_INVALID, "pmg_create: max_episode_steps must be >= 1");
    pmg_dims dims;
    int nb = 0;
    if (fill_dims(cfg, &dims, &nb) != 0)
        return fail(nullptr, PMG_E_INVALID, "pmg_create: unsupported task %d / num_block %d", cfg->task, cfg->num_block);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, PMG_E_DEVICE, "pmg_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, PMG_E_INVALID, "pmg_create: device %d out of range (%d visible)", cfg->device, ndev);
    pmg_env* e = new pmg_env();
    e->cfg = *cfg;
    e->dims = dims;
    e->nb = nb;
    auto bail = [&](int code) { snprintf(g_create_err, sizeof(g_create_err), "%s", e->err); pmg_destroy(e); return code; };
#define CREATE_TRY(call)                                                                         \
    do {                                                                                         \
        hipError_t rc_ = (call);                                                                 \
        if (rc_ != hipSuccess) { fail(e, PMG_E_DEVICE, "%s -> %s", #call, hipGetErrorString(rc_)); return bail(PMG_E_DEVICE); } \
    } while (0)
    CREATE_TRY(hipSetDevice(cfg->device));
    CREATE_TRY(hipStreamCreate(&e->stream));
    CREATE_TRY(hipStreamCreate(&e->side));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    size_t N = (size_t)cfg->num_envs;
    fill_params(e);
    CREATE_TRY(hipMalloc((void**)&e->P.hot, N * pmg::HOT_DIM * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&e->P.cold, N * pmg::COLD_DIM * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&e->P.goal, N * pmg::GOAL_DIM * sizeof(float)));
    if (cfg->use_curriculum) CREATE_TRY(hipMalloc((void**)&e->P.curr, N * pmg::CURR_DIM * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&e->P.blocks, N * pmg::BLOCK_DIM * (nb ? nb : 1) * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&e->P.rng, N * 625 * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc((void**)&e->P.out, N * dims.packed_dim * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&e->P.sched, (4 + 3 * N + 3 * ((N + pmg::PLAN_THREADS - 1) / pmg::PLAN_THREADS)) * sizeof(int)));
    if (const char* pk = getenv("PMG_PACKED")) e->packed = atoi(pk) != 0;
    if (const char* tw = getenv("PMG_REACH_TWO_WAVES")) e->two_wave = atoi(tw) != 0;
    {   /* 1.5 wavefronts per SIMD of this device: how many one-env wavefronts the plan may add to a step */
        int cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || cus <= 0) cus = 256;
        e->P.wave_budget = 6 * cus;
        /* measured on chest_push-4 / chest_pick_and_place-4 x 4096 (tools/chest_exp.sh): 6.5 cm / 6.4 cm put 40-65 % of a
         * random-policy batch on the full-layout list (0.337 M); 5 cm / 2 cm: 20 %, no redo, 0.388 / 0.416 M */
        /* block_stack-4 (tools/near_exp.sh): 6.5 cm puts up to 8 % of the batch on the full-store list, 4.5 cm 1.5 %, no redo
         * either way: 0.549 -> 0.566 M; the one-object tasks are insensitive (0.04 starts to redo) */
        e->P.near_r = e->P.chest >= 0 ? 0.05f : (e->nb > 1 ? 0.045f : 0.065f);
        e->P.chest_reach = 0.02f;
        if (const char* nr = getenv("PMG_NEAR_R")) e->P.near_r = (float)atof(nr);   /* (tuning experiments) */
        if (const char* cr = getenv("PMG_CHEST_REACH")) e->P.chest_reach = (float)atof(cr);
        e->P.list0_prio = -1;
        /* fingers-down class of the one-object tasks: to the one-env list for pick_and_place, packed for push / slide -- by task,
         * never by batch statistics (results must not depend on how a job is sharded; EnvParams::fd_div) */
        e->P.fd_div = e->cfg.task == PMG_TASK_PICK_AND_PLACE ? -1 : 0;
        if (const char* fd = getenv("PMG_FD_DIV")) e->P.fd_div = atoi(fd);
        if (const char* wb = getenv("PMG_WAVE_BUDGET")) e->P.wave_budget = atoi(wb);
        e->P.env_cycles = nullptr;
        /* Longest first on the fast-path list of the many-body tasks (4096 one-env wavefronts on 2048 slots run in two
         * rounds: a long wavefront that starts in the second one is the tail of the step).  The predictor is the env's own
         * wavefront time in the PREVIOUS step (cycles / 64, kept per env: 8 bytes); envs beyond the threshold lead the list
         * (plan_class).  Thresholds by measurement (PMG_LPT_CYCLES sweep 0 / 60 / 70 / 85 / 100 k): chest_push-4 0.42 -> 0.46 M
         * and chest_pick_and_place-4 0.57 -> 0.58 M at 85 k / 70 k, block_stack-4 0.79 -> 0.80 M at 60 k; block_rearrange
         * loses (0.70 -> 0.61 M: there the fingers-down grouping it replaces IS the better order) and keeps it off.
         * Results do not depend on the order of a launch list, only the schedule does */
        /* Round 6: the threshold is no longer a shader-clock count tuned per task on one part at one batch size (85 000 / 70 000 / 60 000):
         * the plan derives it every step as the cycle count above which the slowest lpt_permille / 1000 of the fast-path envs lay in
         * the last step (pmg_k_plan_count's histogram, pmg_k_plan_scatter).  PMG_LPT_PERMILLE overrides the share, PMG_LPT_CYCLES > 0
         * forces a constant, PMG_LPT_CYCLES=0 switches the order off */
        e->P.lpt_thresh = 0;
        /* share of the list that leads it, measured (profiles/r06_lpt_quantile_sweep.txt): chest tasks flat between 350 and 450 (0.403 /
         * 0.545 M, the tuned constants' 0.402 / 0.547 M), block_stack best at 450 (0.804 M; constant 60 000: 0.818 M), block_rearrange loses
         * with any (0.69 -> 0.60 M: its fingers-down grouping is the better order) and keeps the rule off */
        e->P.lpt_permille = dims.num_envs < 4096 ? 0 : ((e->cfg.task == PMG_TASK_CHEST_PUSH || e->cfg.task == PMG_TASK_CHEST_PICK_AND_PLACE) ? PMG_LPT_PERMILLE_DEFAULT :
                                                        (e->cfg.task == PMG_TASK_BLOCK_STACK ? PMG_LPT_PERMILLE_DEFAULT + 50 : 0));
        e->P.lpt_parity = 0;
        e->P.lpt_state = nullptr;
        if (const char* lq = getenv("PMG_LPT_PERMILLE")) e->P.lpt_permille = atoi(lq);
        if (const char* lt = getenv("PMG_LPT_CYCLES")) { e->P.lpt_thresh = atoi(lt); if (e->P.lpt_thresh <= 0) e->P.lpt_permille = 0; }
        const char* ec = getenv("PMG_ENV_CYCLES");
        if (e->nb > 1 && e->P.lpt_permille > 0 && e->P.lpt_thresh <= 0) {
            CREATE_TRY(hipMalloc((void**)&e->P.lpt_state, (2 + pmg::LPT_BINS) * sizeof(int)));
            CREATE_TRY(hipMemset(e->P.lpt_state, 0, (2 + pmg::LPT_BINS) * sizeof(int)));
        }
        if ((ec && atoi(ec) != 0) || ((e->P.lpt_thresh > 0 || e->P.lpt_permille > 0) && e->nb > 1)) {
            CREATE_TRY(hipMalloc((void**)&e->P.env_cycles, 2 * N * sizeof(int)));
            CREATE_TRY(hipMemset(e->P.env_cycles, 0, 2 * N * sizeof(int)));
        }
        if (const char* pr = getenv("PMG_LIST0_PRIO")) e->P.list0_prio = atoi(pr);
    }
    CREATE_TRY(hipMalloc((void**)&e->d_actions, N * dims.action_dim * sizeof(float)));
    CREATE_TRY(hipMalloc((void**)&e->d_mask, N));
    CREATE_TRY(hipHostMalloc((void**)&e->h_packed, N * dims.packed_dim * sizeof(float)));
    CREATE_TRY(hipHostMalloc((void**)&e->h_actions, N * dims.action_dim * sizeof(float)));
    for (int i = 0; i < EVENT_POOL; i++) { CREATE_TRY(hipEventCreate(&e->ev_a[i])); CREATE_TRY(hipEventCreate(&e->ev_b[i])); }
    for (int i = 0; i < EVENT_POOL; i++) { CREATE_TRY(hipEventCreate(&e->cv_a[i])); CREATE_TRY(hipEventCreate(&e->cv_b[i])); }
    /* initial state: rest pose kuka.py:27, blocks parked below the floor */
    {
        static const float rest0[7] = {0.f, -0.5592432f, 0.f, 1.733180f, 0.f, -0.8501557f, 0.f};
        std::vector<float> cold(N * pmg::COLD_DIM, 0.f), blk(N * pmg::BLOCK_DIM * (nb ? nb : 1), 0.f);
        for (size_t i = 0; i < N; i++) {
            memcpy(&cold[i * pmg::COLD_DIM], rest0, sizeof(rest0));
            for (int b = 0; b < nb; b++) { float* o = &blk[(i * nb + b) * pmg::BLOCK_DIM]; o[2] = -3.f; o[6] = 1.f; }
        }
        CREATE_TRY(hipMemcpy(e->P.cold, cold.data(), cold.size() * sizeof(float), hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(e->P.blocks, blk.data(), blk.size() * sizeof(float), hipMemcpyHostToDevice));
        {   /* identity launch schedule: all envs in the contact-prone list */
            std::vector<int> sc(4 + 3 * N + 3 * ((N + pmg::PLAN_THREADS - 1) / pmg::PLAN_THREADS), 0);   /* (+ the plan's counts and its promotion flag) */
            sc[0] = (int)N;
            for (size_t i = 0; i < N; i++) sc[2 + i] = (int)i;
            CREATE_TRY(hipMemcpy(e->P.sched, sc.data(), sc.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMemset(e->P.hot, 0, N * pmg::HOT_DIM * sizeof(float)));
        CREATE_TRY(hipMemset(e->P.goal, 0, N * pmg::GOAL_DIM * sizeof(float)));
        if (e->P.curr) { /* start with the easiest goal as the only possible one (kuka_multi_step_base_env.py:133) */
            std::vector<float> cs(N * pmg::CURR_DIM, 0.f);
            const int NC = e->P.chest >= 0 ? 6 : 5;   /* row: prob[NC] | generated[NC] | goal_step */
            for (size_t i = 0; i < N; i++) { cs[i * pmg::CURR_DIM] = 1.f; cs[i * pmg::CURR_DIM + 2 * NC] = 50.f; }
            CREATE_TRY(hipMemcpy(e->P.curr, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMemset(e->P.out, 0, N * dims.packed_dim * sizeof(float)));
    }
    if (upload_seeds(e) != PMG_OK) return bail(PMG_E_DEVICE);
    {   /* the tuning / experiment switches this library reads from the environment change its schedule (never its results' meaning):
         * a stray one in a user's shell -- PMG_PACKED=0 halves the throughput -- must not go unnoticed: ONE line on stderr per handle */
        static const char* const kSwitches[] = {"PMG_PACKED", "PMG_REACH_TWO_WAVES", "PMG_NEAR_R", "PMG_CHEST_REACH", "PMG_FD_DIV", "PMG_WAVE_BUDGET",
                                                "PMG_LPT_CYCLES", "PMG_LPT_PERMILLE", "PMG_ENV_CYCLES", "PMG_LIST0_PRIO", "PMG_LIST0_FIRST", "PMG_PLAN_TWO_PASS",
                                                "PMG_REWARD_GENERIC"};
        std::string active;
        for (const char* name : kSwitches)
            if (const char* v = getenv(name)) { active += active.empty() ? "" : " "; active += name; active += "="; active += v; }
        if (!active.empty()) fprintf(stderr, "[libpmg_hip] environment overrides active for this handle: %s\n", active.c_str());
    }
    *out = e;
    return PMG_OK;
}

void pmg_destroy(pmg_env* e)
{
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);   /* an overlapped all-gather in flight reads out2[] through e->comm */
    if (e->comm) ncclCommDestroy(e->comm);
    (void)hipFree(e->P.hot); (void)hipFree(e->P.cold); (void)hipFree(e->P.goal); (void)hipFree(e->P.curr); (void)hipFree(e->P.blocks); (void)hipFree(e->P.rng); if (e->out2[1]) { (void)hipFree(e->out2[0]); (void)hipFree(e->out2[1]); } else (void)hipFree(e->P.out); (void)hipFree(e->P.sched); if (e->P.env_cycles) (void)hipFree(e->P.env_cycles); if (e->P.lpt_state) (void)hipFree(e->P.lpt_state);
    (void)hipFree(e->d_actions); (void)hipFree(e->d_mask);
    (void)hipFree(e->d_rw_ag); (void)hipFree(e->d_rw_dg); (void)hipFree(e->d_rw_r); (void)hipFree(e->d_rw_ok);
    if (e->h_packed) (void)hipHostFree(e->h_packed);
    if (e->h_actions) (void)hipHostFree(e->h_actions);
    for (int i = 0; i < EVENT_POOL; i++) { if (e->ev_a[i]) (void)hipEventDestroy(e->ev_a[i]); if (e->ev_b[i]) (void)hipEventDestroy(e->ev_b[i]); }
    for (int i = 0; i < EVENT_POOL; i++) { if (e->cv_a[i]) (void)hipEventDestroy(e->cv_a[i]); if (e->cv_b[i]) (void)hipEventDestroy(e->cv_b[i]); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->side) { (void)hipStreamSynchronize(e->side); (void)hipStreamDestroy(e->side); }
    if (e->comm_stream) { (void)hipStreamSynchronize(e->comm_stream); (void)hipStreamDestroy(e->comm_stream); }
    for (int b = 0; b < 2; b++) { if (e->ev_rows[b]) (void)hipEventDestroy(e->ev_rows[b]); if (e->ev_gdone[b]) (void)hipEventDestroy(e->ev_gdone[b]); }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int pmg_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return 0;
    return ndev;
}

int pmg_get_dims(const pmg_env* e, pmg_dims* out)
{
    if (!e || !out) return PMG_E_INVALID;
    *out = e->dims;
    return PMG_OK;
}

int pmg_seed(pmg_env* e, uint64_t base, uint64_t stride)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    e->cfg.seed_base = base;
    e->cfg.seed_stride = stride;
    return upload_seeds(e);
}

/* Overlapped all-gather: a gather of the CURRENT row buffer may still be reading it on the communication stream (the step that
 * wrote it was gathered, the next step has not been queued yet).  Whatever writes rows into that buffer between two steps -- a
 * reset, set_sub_goal, the reset of nobody behind pmg_set_state -- goes behind that gather; the gathered table of step t then
 * holds step t's rows and nothing of the reset that followed it (bench.py --lockstep resets right behind the gather). */
static int rows_writer_waits_for_gather(pmg_env* e)
{
    if (e->overlap && e->gpending[e->out_phase]) {
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_gdone[e->out_phase], 0));
        e->gpending[e->out_phase] = false;
    }
    return PMG_OK;
}

int pmg_reset_device(pmg_env* e, const uint8_t* d_mask)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (int rc = rows_writer_waits_for_gather(e)) return rc;
    HIP_TRY(e, pmg_launch_reset(e->P, d_mask, e->stream));
    if (!d_mask) e->ever_reset = true;
    return PMG_OK;
}

int pmg_reset_done_device(pmg_env* e)
{
    if (!e) return PMG_E_INVALID;
    if (!e->ever_reset) return fail(e, PMG_E_STATE, "pmg_reset_done_device: reset() must be called (for all envs) first");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (int rc = rows_writer_waits_for_gather(e)) return rc;
    HIP_TRY(e, pmg_launch_reset(e->P, nullptr, e->stream, 1));
    return PMG_OK;
}

int pmg_step_device(pmg_env* e, const float* d_actions)
{
    if (!e || !d_actions) return PMG_E_INVALID;
    if (!e->ever_reset) return fail(e, PMG_E_STATE, "pmg_step: reset() must be called (for all envs) before the first step()");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (e->overlap) {
        /* this step writes the other row buffer; the all-gather that last read it (two steps ago) must be through */
        const int b = e->out_phase ^ 1;
        if (e->gpending[b]) { HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_gdone[b], 0)); e->gpending[b] = false; }
        e->out_phase = b;
        e->P.out = e->out2[b];
    }
    const bool timed = (e->step_count++ % e->ev_every) == 0;
    if (timed && e->ev_n == EVENT_POOL) drain_events(e);
    int i = timed ? e->ev_n++ : 0;
    e->P.lpt_parity = (int)(e->plans++ & 1);
    HIP_TRY(e, pmg_launch_plan(e->P, d_actions, e->stream)); /* launch-order plan (13 us), outside the step-kernel timer */
    int mode = e->packed;
    /* reach: steps that have contact-prone envs run the two-wavefront kernel while the contact-free list is at most one
     * wavefront per SIMD (pmg_kernels.hip); mode 2 launches both kernels and the DEVICE picks one by the plan's count */
    if (e->packed && e->nb == 0 && !e->cfg.joint_control && e->two_wave && e->dims.num_envs <= (e->P.wave_budget / 6) * 16) mode = 2;
    if (timed) HIP_TRY(e, hipEventRecord(e->ev_a[i], e->stream));
    HIP_TRY(e, pmg_launch_step(e->P, d_actions, e->stream, mode, e->side, e->ev_fork, e->ev_join));
    if (timed) HIP_TRY(e, hipEventRecord(e->ev_b[i], e->stream));
    return PMG_OK;
}

int pmg_read_outputs(pmg_env* e, float* obs, float* pol, float* ag, float* dg, float* reward, uint8_t* ga, uint8_t* done)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    size_t bytes = (size_t)e->dims.num_envs * e->dims.packed_dim * sizeof(float);
    HIP_TRY(e, hipMemcpyAsync(e->h_packed, e->P.out, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    unpack(e, e->h_packed, obs, pol, ag, dg, reward, ga, done);
    return PMG_OK;
}

int pmg_reset(pmg_env* e, const uint8_t* mask, float* obs, float* pol, float* ag, float* dg)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const uint8_t* dm = nullptr;
    if (mask) {
        if (!e->ever_reset) {
            for (int i = 0; i < e->dims.num_envs; i++)
                if (!mask[i]) return fail(e, PMG_E_STATE, "pmg_reset: the first reset must cover every env");
        }
        HIP_TRY(e, hipMemcpyAsync(e->d_mask, mask, (size_t)e->dims.num_envs, hipMemcpyHostToDevice, e->stream));
        dm = e->d_mask;
    }
    int rc = pmg_reset_device(e, dm);
    if (rc != PMG_OK) return rc;
    e->ever_reset = true;
    return pmg_read_outputs(e, obs, pol, ag, dg, nullptr, nullptr, nullptr);
}

int pmg_step(pmg_env* e, const float* actions, float* obs, float* pol, float* ag, float* dg, float* reward, uint8_t* ga, uint8_t* done)
{
    if (!e || !actions) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    size_t n = (size_t)e->dims.num_envs * e->dims.action_dim;
    for (size_t i = 0; i < n; i++) {
        /* kuka.py:168 assert action_space.contains(a): Box(-1, 1) */
        if (!(actions[i] >= -1.f && actions[i] <= 1.f)) return fail(e, PMG_E_INVALID, "pmg_step: action[%zu] = %g is outside [-1, 1]", i, (double)actions[i]);
        e->h_actions[i] = actions[i];
    }
    HIP_TRY(e, hipMemcpyAsync(e->d_actions, e->h_actions, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    int rc = pmg_step_device(e, e->d_actions);
    if (rc != PMG_OK) return rc;
    return pmg_read_outputs(e, obs, pol, ag, dg, reward, ga, done);
}

int pmg_device_ptr(pmg_env* e, int which, void** d_ptr)
{
    if (!e || !d_ptr) return PMG_E_INVALID;
    switch (which) {
    case PMG_BUF_PACKED: *d_ptr = e->P.out; return PMG_OK;
    case PMG_BUF_STATE: *d_ptr = e->P.hot; return PMG_OK;
    case PMG_BUF_SCHED: *d_ptr = e->P.sched; return PMG_OK;
    case PMG_BUF_ENV_CYCLES:
        if (!e->P.env_cycles) return fail(e, PMG_E_INVALID, "pmg_device_ptr: PMG_BUF_ENV_CYCLES is kept only with PMG_ENV_CYCLES=1 in the environment at pmg_create, or with the longest-first order on (PMG_LPT_CYCLES)");
        *d_ptr = e->P.env_cycles; return PMG_OK;
    default: return fail(e, PMG_E_INVALID, "pmg_device_ptr: buffer %d is not a device buffer (use PMG_BUF_PACKED + pmg_dims offsets)", which);
    }
}
int pmg_stream(pmg_env* e, void** s)
{
    if (!e || !s) return PMG_E_INVALID;
    *s = (void*)e->stream;
    return PMG_OK;
}
int pmg_sync(pmg_env* e)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->comm_stream) HIP_TRY(e, hipStreamSynchronize(e->comm_stream));   /* (an overlapped all-gather in flight) */
    return PMG_OK;
}

int pmg_compute_reward_device(pmg_env* e, const float* d_ag, const float* d_dg, int64_t batch, float* d_r, uint8_t* d_ok)
{
    if (!e || !d_ag || !d_dg || batch < 0) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, pmg_launch_reward(d_ag, d_dg, batch, e->dims.goal_dim, e->cfg.distance_threshold, e->cfg.binary_reward, d_r, d_ok, e->stream));
    return PMG_OK;
}

int pmg_compute_reward(pmg_env* e, const float* ag, const float* dg, int64_t batch, float* r, uint8_t* ok)
{
    if (!e || !ag || !dg || batch < 0) return PMG_E_INVALID;
    if (batch == 0) return PMG_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    int G = e->dims.goal_dim;
    if (batch > e->rw_cap) {
        (void)hipFree(e->d_rw_ag); (void)hipFree(e->d_rw_dg); (void)hipFree(e->d_rw_r); (void)hipFree(e->d_rw_ok);
        e->d_rw_ag = e->d_rw_dg = e->d_rw_r = nullptr; e->d_rw_ok = nullptr; e->rw_cap = 0;
        HIP_TRY(e, hipMalloc((void**)&e->d_rw_ag, (size_t)batch * G * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&e->d_rw_dg, (size_t)batch * G * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&e->d_rw_r, (size_t)batch * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&e->d_rw_ok, (size_t)batch));
        e->rw_cap = batch;
    }
    HIP_TRY(e, hipMemcpyAsync(e->d_rw_ag, ag, (size_t)batch * G * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(e->d_rw_dg, dg, (size_t)batch * G * sizeof(float), hipMemcpyHostToDevice, e->stream));
    int rc = pmg_compute_reward_device(e, e->d_rw_ag, e->d_rw_dg, batch, e->d_rw_r, e->d_rw_ok);
    if (rc != PMG_OK) return rc;
    if (r) HIP_TRY(e, hipMemcpyAsync(r, e->d_rw_r, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (ok) HIP_TRY(e, hipMemcpyAsync(ok, e->d_rw_ok, (size_t)batch, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return PMG_OK;
}

/* state row = hot(32) | cold(16) | goal(16) | blocks(13 nb)    */
int pmg_get_state(pmg_env* e, float* state)
{
    if (!e || !state) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    size_t N = (size_t)e->dims.num_envs;
    int S = e->dims.state_dim, nbd = pmg::BLOCK_DIM * e->nb;
    std::vector<float> hot(N * pmg::HOT_DIM), cold(N * pmg::COLD_DIM), goal(N * pmg::GOAL_DIM), blk(N * (nbd ? nbd : 1));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(hot.data(), e->P.hot, hot.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(cold.data(), e->P.cold, cold.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(goal.data(), e->P.goal, goal.size() * sizeof(float), hipMemcpyDeviceToHost));
    if (nbd) HIP_TRY(e, hipMemcpy(blk.data(), e->P.blocks, N * nbd * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        float* s = state + i * S;
        memcpy(s, &hot[i * pmg::HOT_DIM], sizeof(float) * pmg::HOT_DIM);
        memcpy(s + 32, &cold[i * pmg::COLD_DIM], sizeof(float) * pmg::COLD_DIM);
        memcpy(s + 48, &goal[i * pmg::GOAL_DIM], sizeof(float) * pmg::GOAL_DIM);
        if (nbd) memcpy(s + 64, &blk[i * nbd], sizeof(float) * nbd);
    }
    if (e->P.curr) {
        std::vector<float> cs(N * pmg::CURR_DIM);
        HIP_TRY(e, hipMemcpy(cs.data(), e->P.curr, cs.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; i++) memcpy(state + i * S + 64 + nbd, &cs[i * pmg::CURR_DIM], sizeof(float) * pmg::CURR_DIM);
    }
    return PMG_OK;
}

int pmg_set_state(pmg_env* e, const float* state)
{
    if (!e || !state) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    size_t N = (size_t)e->dims.num_envs;
    int S = e->dims.state_dim, nbd = pmg::BLOCK_DIM * e->nb;
    std::vector<float> hot(N * pmg::HOT_DIM), cold(N * pmg::COLD_DIM), goal(N * pmg::GOAL_DIM), blk(N * (nbd ? nbd : 1));
    for (size_t i = 0; i < N; i++) {
        const float* s = state + i * S;
        memcpy(&hot[i * pmg::HOT_DIM], s, sizeof(float) * pmg::HOT_DIM);
        memcpy(&cold[i * pmg::COLD_DIM], s + 32, sizeof(float) * pmg::COLD_DIM);
        memcpy(&goal[i * pmg::GOAL_DIM], s + 48, sizeof(float) * pmg::GOAL_DIM);
        if (nbd) memcpy(&blk[i * nbd], s + 64, sizeof(float) * nbd);
    }
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(e->P.hot, hot.data(), hot.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->P.cold, cold.data(), cold.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->P.goal, goal.data(), goal.size() * sizeof(float), hipMemcpyHostToDevice));
    if (nbd) HIP_TRY(e, hipMemcpy(e->P.blocks, blk.data(), N * nbd * sizeof(float), hipMemcpyHostToDevice));
    if (e->P.curr) {
        std::vector<float> cs(N * pmg::CURR_DIM);
        for (size_t i = 0; i < N; i++) memcpy(&cs[i * pmg::CURR_DIM], state + i * S + 64 + nbd, sizeof(float) * pmg::CURR_DIM);
        HIP_TRY(e, hipMemcpy(e->P.curr, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    e->ever_reset = true;
    /* the packed rows still show the previous state: a reset of NOBODY re-derives every env's observation and goal */
    HIP_TRY(e, hipMemsetAsync(e->d_mask, 0, N, e->stream));
    if (int rc = rows_writer_waits_for_gather(e)) return rc;
    HIP_TRY(e, pmg_launch_reset(e->P, e->d_mask, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return PMG_OK;
}

int pmg_get_rng(pmg_env* e, uint32_t* words)
{
    if (!e || !words) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(words, e->P.rng, (size_t)e->dims.num_envs * 625 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PMG_OK;
}
int pmg_set_rng(pmg_env* e, const uint32_t* words)
{
    if (!e || !words) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    for (int i = 0; i < e->dims.num_envs; i++)
        if (words[(size_t)i * 625 + 624] > 624u) return fail(e, PMG_E_INVALID, "pmg_set_rng: env %d has cursor %u > 624", i, words[(size_t)i * 625 + 624]);
    HIP_TRY(e, hipMemcpy(e->P.rng, words, (size_t)e->dims.num_envs * 625 * sizeof(uint32_t), hipMemcpyHostToDevice));
    return PMG_OK;
}

int pmg_set_goal(pmg_env* e, const uint8_t* mask, const float* goals)
{
    if (!e || !goals) return PMG_E_INVALID;
    if (e->P.chest >= 0) return fail(e, PMG_E_INVALID, "pmg_set_goal: the chest tasks have no static target");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    size_t N = (size_t)e->dims.num_envs;
    int G = e->dims.goal_dim;
    std::vector<float> goal(N * pmg::GOAL_DIM);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(goal.data(), e->P.goal, goal.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++)
        if (!mask || mask[i]) memcpy(&goal[i * pmg::GOAL_DIM], goals + i * G, sizeof(float) * (G < 15 ? G : 15)); /* static targets only */
    HIP_TRY(e, hipMemcpy(e->P.goal, goal.data(), goal.size() * sizeof(float), hipMemcpyHostToDevice));
    return PMG_OK;
}

int pmg_set_sub_goal(pmg_env* e, const uint8_t* mask, int32_t sub_goal_ind)
{
    if (!e) return PMG_E_INVALID;
    if (!e->cfg.task_decomposition) return fail(e, PMG_E_STATE, "pmg_set_sub_goal: the handle was created without task_decomposition");
    int steps = e->cfg.grip_informed_goal ? 2 * e->nb : e->nb; /* kuka_multi_step_envs.py:13-17 */
    if (e->P.chest >= 0) steps = (e->cfg.grip_informed_goal ? e->nb * (e->P.grasping ? 3 : 2) : e->nb) + 1; /* :238-242, 388-392 */
    if (sub_goal_ind < -1 || sub_goal_ind >= steps) return fail(e, PMG_E_INVALID, "pmg_set_sub_goal: index %d out of range [-1, %d)", sub_goal_ind, steps);
    if (!e->ever_reset) return fail(e, PMG_E_STATE, "pmg_set_sub_goal: reset first");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const unsigned char* dm = nullptr;
    if (mask) {
        HIP_TRY(e, hipMemcpyAsync(e->d_mask, mask, (size_t)e->dims.num_envs, hipMemcpyHostToDevice, e->stream));
        dm = e->d_mask;
    }
    if (int rc = rows_writer_waits_for_gather(e)) return rc;
    HIP_TRY(e, pmg_launch_sub_goal(e->P, dm, sub_goal_ind < 0 ? steps - 1 : sub_goal_ind, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return PMG_OK;
}

int pmg_curriculum_update(pmg_env* e, int32_t enabled)
{
    if (!e) return PMG_E_INVALID;
    if (!e->cfg.use_curriculum) return fail(e, PMG_E_STATE, "pmg_curriculum_update: the handle was created without use_curriculum");
    e->P.curriculum_update = enabled != 0;
    return PMG_OK;
}

int pmg_curriculum_read(pmg_env* e, int32_t* level, int32_t* goal_step, float* prob, float* generated)
{
    if (!e) return PMG_E_INVALID;
    if (!e->cfg.use_curriculum) return fail(e, PMG_E_STATE, "pmg_curriculum_read: the handle was created without use_curriculum");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    size_t N = (size_t)e->dims.num_envs;
    std::vector<float> cs(N * pmg::CURR_DIM), cold(N * pmg::COLD_DIM);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(cs.data(), e->P.curr, cs.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(cold.data(), e->P.cold, cold.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) {
        if (level) level[i] = (int32_t)cold[i * pmg::COLD_DIM + 7];
        const int ncur = e->P.chest >= 0 ? e->nb + 1 : e->nb, NC = e->P.chest >= 0 ? 6 : 5;
        if (goal_step) goal_step[i] = (int32_t)cs[i * pmg::CURR_DIM + 2 * NC];
        for (int b = 0; b < ncur; b++) {
            if (prob) prob[i * ncur + b] = cs[i * pmg::CURR_DIM + b];
            if (generated) generated[i * ncur + b] = cs[i * pmg::CURR_DIM + NC + b];
        }
    }
    return PMG_OK;
}

int pmg_comm_unique_id(uint8_t id[128])
{
    ncclUniqueId u;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    if (ncclGetUniqueId(&u) != ncclSuccess) return PMG_E_COMM;
    memcpy(id, &u, 128);
    return PMG_OK;
}
int pmg_comm_init(pmg_env* e, int rank, int nranks, const uint8_t id[128])
{
    if (!e || !id || nranks < 1 || rank < 0 || rank >= nranks) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t rc = ncclCommInitRank(&e->comm, nranks, u, rank);
    if (rc != ncclSuccess) return fail(e, PMG_E_COMM, "ncclCommInitRank -> %s", ncclGetErrorString(rc));
    e->rank = rank;
    e->nranks = nranks;
    return PMG_OK;
}
int pmg_allgather_packed(pmg_env* e, float* d_gathered)
{
    if (!e || !d_gathered) return PMG_E_INVALID;
    if (!e->comm) return fail(e, PMG_E_STATE, "pmg_allgather_packed: pmg_comm_init was not called");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    size_t count = (size_t)e->dims.num_envs * e->dims.packed_dim;
    if (e->cv_n == EVENT_POOL) drain_comm_events(e);
    /* the event pair counts only once both are recorded: a failed all-gather must not leave a new cv_a paired with a stale cv_b */
    const int i = e->cv_n;
    HIP_TRY(e, hipEventRecord(e->cv_a[i], e->stream));
    ncclResult_t rc = ncclAllGather(e->P.out, d_gathered, count, ncclFloat, e->comm, e->stream);
    if (rc != ncclSuccess) return fail(e, PMG_E_COMM, "ncclAllGather -> %s", ncclGetErrorString(rc));
    HIP_TRY(e, hipEventRecord(e->cv_b[i], e->stream));
    e->cv_n = i + 1;
    return PMG_OK;
}

int pmg_comm_overlap(pmg_env* e, int32_t enabled)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (!enabled) {
        if (e->overlap) {
            if (e->comm_stream) HIP_TRY(e, hipStreamSynchronize(e->comm_stream));
            e->gpending[0] = e->gpending[1] = false;
            e->last_gather = -1;
        }
        e->overlap = false;       /* (the rows stay where the last step wrote them: P.out keeps pointing at that buffer) */
        return PMG_OK;
    }
    if (!(e->out2[1] && e->comm_stream)) {
        /* everything into locals first: a failure half-way must not leave a handle that looks initialised (out2[0] set, the second
         * buffer / the stream / the events null) to the next call */
        const size_t bytes = (size_t)e->dims.num_envs * e->dims.packed_dim * sizeof(float);
        float* second = nullptr;
        hipStream_t cs = nullptr;
        hipEvent_t rows[2] = {nullptr, nullptr}, gdone[2] = {nullptr, nullptr};
        hipError_t rc = hipMalloc((void**)&second, bytes);
        if (rc == hipSuccess) rc = hipMemcpyAsync(second, e->P.out, bytes, hipMemcpyDeviceToDevice, e->stream);
        if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        for (int b = 0; b < 2 && rc == hipSuccess; b++) {
            rc = hipEventCreateWithFlags(&rows[b], hipEventDisableTiming);
            if (rc == hipSuccess) rc = hipEventCreateWithFlags(&gdone[b], hipEventDisableTiming);
        }
        if (rc != hipSuccess) {
            (void)hipStreamSynchronize(e->stream);
            for (int b = 0; b < 2; b++) { if (rows[b]) (void)hipEventDestroy(rows[b]); if (gdone[b]) (void)hipEventDestroy(gdone[b]); }
            if (cs) (void)hipStreamDestroy(cs);
            if (second) (void)hipFree(second);
            return fail(e, second ? PMG_E_DEVICE : PMG_E_NOMEM, "pmg_comm_overlap: %s", hipGetErrorString(rc));
        }
        e->out2[0] = e->P.out; e->out2[1] = second;
        e->comm_stream = cs;
        for (int b = 0; b < 2; b++) { e->ev_rows[b] = rows[b]; e->ev_gdone[b] = gdone[b]; }
        e->out_phase = 0;
    }
    e->overlap = true;
    return PMG_OK;
}
int pmg_allgather_packed_async(pmg_env* e, float* d_gathered)
{
    if (!e || !d_gathered) return PMG_E_INVALID;
    if (!e->comm) return fail(e, PMG_E_STATE, "pmg_allgather_packed_async: pmg_comm_init was not called");
    if (!e->overlap) return fail(e, PMG_E_STATE, "pmg_allgather_packed_async: pmg_comm_overlap(env, 1) first");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const size_t count = (size_t)e->dims.num_envs * e->dims.packed_dim;
    const int b = e->out_phase;
    if (e->cv_n == EVENT_POOL) drain_comm_events(e);
    const int i = e->cv_n;
    /* rows of this step (and of the masked resets behind it) are complete on the step's stream -> the communication stream
     * may read them; nothing else of the step's stream waits for the transfer */
    HIP_TRY(e, hipEventRecord(e->ev_rows[b], e->stream));
    HIP_TRY(e, hipStreamWaitEvent(e->comm_stream, e->ev_rows[b], 0));
    HIP_TRY(e, hipEventRecord(e->cv_a[i], e->comm_stream));
    ncclResult_t rc = ncclAllGather(e->out2[b], d_gathered, count, ncclFloat, e->comm, e->comm_stream);
    if (rc != ncclSuccess) return fail(e, PMG_E_COMM, "ncclAllGather -> %s", ncclGetErrorString(rc));
    HIP_TRY(e, hipEventRecord(e->cv_b[i], e->comm_stream));
    HIP_TRY(e, hipEventRecord(e->ev_gdone[b], e->comm_stream));
    e->cv_n = i + 1;
    e->gpending[b] = true;
    e->last_gather = b;
    return PMG_OK;
}
int pmg_allgather_wait(pmg_env* e, int32_t host)
{
    if (!e) return PMG_E_INVALID;
    if (e->last_gather < 0) return PMG_OK;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (host) HIP_TRY(e, hipEventSynchronize(e->ev_gdone[e->last_gather]));
    else HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_gdone[e->last_gather], 0));
    return PMG_OK;
}

int pmg_device_alloc(pmg_env* e, uint64_t bytes, void** d_ptr)
{
    if (!e || !d_ptr) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) return fail(e, PMG_E_NOMEM, "pmg_device_alloc: hipMalloc(%llu) failed", (unsigned long long)bytes);
    return PMG_OK;
}
int pmg_device_free(pmg_env* e, void* d_ptr)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipFree(d_ptr));
    return PMG_OK;
}
int pmg_upload(pmg_env* e, void* d_dst, const void* h_src, uint64_t bytes)
{
    if (!e || !d_dst || !h_src) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return PMG_OK;
}
int pmg_download(pmg_env* e, void* h_dst, const void* d_src, uint64_t bytes)
{
    if (!e || !h_dst || !d_src) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return PMG_OK;
}

int pmg_timing_reset(pmg_env* e)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    e->ev_n = 0;
    e->step_count = 0;
    e->ev_ms = e->ev_min = e->ev_max = 0.0;
    e->ev_launches = 0;
    e->cv_n = 0;
    e->cv_ms = e->cv_max = 0.0;
    e->cv_launches = 0;
    return PMG_OK;
}
int pmg_timing_every(pmg_env* e, int n)
{
    if (!e || n < 1) return PMG_E_INVALID;
    e->ev_every = n;
    e->step_count = 0;
    return PMG_OK;
}
int pmg_timing_stats(pmg_env* e, double* min_ms, double* avg_ms, double* max_ms, int64_t* launches)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    drain_events(e);
    if (min_ms) *min_ms = e->ev_min;
    if (max_ms) *max_ms = e->ev_max;
    if (avg_ms) *avg_ms = e->ev_launches ? e->ev_ms / (double)e->ev_launches : 0.0;
    if (launches) *launches = e->ev_launches;
    return PMG_OK;
}
int pmg_comm_timing(pmg_env* e, double* avg_ms, double* max_ms, int64_t* launches)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    drain_comm_events(e);
    if (avg_ms) *avg_ms = e->cv_launches ? e->cv_ms / (double)e->cv_launches : 0.0;
    if (max_ms) *max_ms = e->cv_max;
    if (launches) *launches = e->cv_launches;
    return PMG_OK;
}
int pmg_timing_read(pmg_env* e, double* avg_ms, int64_t* launches)
{
    if (!e) return PMG_E_INVALID;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    drain_events(e);
    if (avg_ms) *avg_ms = e->ev_launches ? e->ev_ms / (double)e->ev_launches : 0.0;
    if (launches) *launches = e->ev_launches;
    return PMG_OK;
}

}  /* extern "C" */
