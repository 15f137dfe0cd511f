/* pmg_probe.cpp -- TEST INFRASTRUCTURE: runs individual device functions of
 * pybullet_multigoal_gym_amd/csrc/pmg_device.h on the fiber emulator so that unit tests can
 * compare them with the oracle's probes. */
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "pmg_kernels.h"

extern "C" void pmge_probe_dynamics(const float* q, const float* qd, const float* tau, float* qdd, float* minv_out,
                                    float* tip_out)
{
    emu::launch(1, 64, [&]() {
        using namespace pmg;
        int l = wv::lane();
        __shared__ LaneTabStore lcs;
        LaneConst c;
        load_lane_const(lcs, c);
        float ql = l < NJ ? q[l] : 0.f, qdl = l < NJ ? qd[l] : 0.f, tl = l < NJ ? tau[l] : 0.f;
        Kin k;
        fk(c, ql, k);
        float tip[3], Rt[9];
        tip_frame(k, tip, Rt);
        float I10[10], minv[NJ], v[6];
        body_inertia(c, k, I10);
        if (l >= NJ)
            for (int a = 0; a < 10; a++) I10[a] = 0.f;
        mass_inverse(k, I10, minv);
        float h = bias_torque(c, k, I10, qdl);
        float rq = l < NJ ? tl - h : 0.f;
        float acc = 0.f;
        for (int j = 0; j < NJ; j++) acc += minv[j] * wv::bcast(rq, j);
        if (l < NJ) {
            qdd[l] = acc;
            for (int j = 0; j < NJ; j++) minv_out[9 * l + j] = minv[j];
        }
        if (l == 0) {
            for (int a = 0; a < 3; a++) tip_out[a] = tip[a];
            for (int a = 0; a < 9; a++) tip_out[3 + a] = Rt[a];
        }
    });
}

extern "C" int pmge_probe_ik(const float* q, const float* target, float* q_out)
{
    emu::launch(1, 64, [&]() {
        using namespace pmg;
        int l = wv::lane();
        __shared__ LaneTabStore lcs;
        LaneConst c;
        load_lane_const(lcs, c);
        float ql = l < NJ ? q[l] : 0.f;
        float r = ik_solve(c, ql, target);
        if (l < NJ) q_out[l] = r;
    });
    return 0;
}

/* device narrowphase on one pair (plain call: these functions use no cross-lane primitive).  kind 0: box_box_fast
 * (register front end, box_face_clip for partial face overlaps, box_box_resume for edges), 1: cyl_box with A = cylinder
 * (half = r, r, hl), 2: the general box_box alone, 3: box_box_fast with B's rotation known to be the identity.  out: n x 10 floats */
extern "C" int pmge_probe_narrowphase(int kind, const float* ca, const float* Ra, const float* ha, const float* cb, const float* Rb,
                                      const float* hb, float margin, float* out)
{
    alignas(16) static float W[256];
    if (kind == 0) return pmg::box_box_fast(ca, Ra, ha, cb, Rb, hb, margin, out, W);
    if (kind == 2) return pmg::box_box(ca, Ra, ha, cb, Rb, hb, margin, out, W);
    if (kind == 3) return pmg::box_box_fast<true, true>(ca, Ra, ha, cb, Rb, hb, margin, out, W);   /* the reach kernel's instantiation: Rb known to be the identity */   /* the general routine alone (SAT + clipping through the workspace) */
    return pmg::cyl_box(ca, Ra, ha[0], ha[2], cb, Rb, hb[0], hb[1], hb[2], margin, out, W);
}

/* the double-precision repeat of a cylinder pair (cyl_redo64) and its pieces, called directly: the double forward kinematics of a
 * contact body's link, and the repeat itself for chest kind ck (-1: none).  amb_out (may be NULL): the ambiguity of the FLOAT
 * pass over the same pair taken from float poses (ca, Ra | cb, Rb given by the caller), the trigger of the repeat */
extern "C" void pmge_probe_fk64(const float* q9, int body, double* p, double* R) { pmg::fk64_link(q9, body, p, R); }
extern "C" int pmge_probe_cyl_redo64(int ck, int cyl_body, int box_body, int wall, const float* q9, const float* blk0, const float* doorq,
                                     const float* kc, float prad, float phl, float* out)
{
    alignas(16) static float W[256];
    if (ck == 0) return pmg::cyl_redo64<0>(cyl_body, box_body, wall, q9, blk0, doorq, kc, prad, phl, out, W);
    if (ck == 1) return pmg::cyl_redo64<1>(cyl_body, box_body, wall, q9, blk0, doorq, kc, prad, phl, out, W);
    return pmg::cyl_redo64<-1>(cyl_body, box_body, wall, q9, blk0, doorq, kc, prad, phl, out, W);
}
extern "C" int pmge_probe_cyl_amb(const float* ca, const float* Ra, float rad, float hl, const float* cb, const float* Rb, const float* hb, float margin,
                                  float* out, float* amb_out)
{
    alignas(16) static float W[256];
    float amb = 1e30f;
    const int n = pmg::cyl_box(ca, Ra, rad, hl, cb, Rb, hb[0], hb[1], hb[2], margin, out, W, &amb);
    *amb_out = amb;
    return n;
}

/* the launch-order plan in isolation: the single-workgroup plan (two_pass = 0) or the two-pass multi-workgroup plan on
 * the same batch; sched_out = [3 + 3 N] as on the device; returns the plan's promotion flag (< 0: not run).  hot: [N, 32] state rows, blocks: [N, 13 nb], actions [N, adim] */
extern "C" int pmge_probe_plan(int n_envs, int nb, const float* hot, const float* blocks, const float* actions, int adim,
                               int wave_budget, int two_pass, int* sched_out)
{
    using namespace pmg;
    EnvParams P;
    memset(&P, 0, sizeof(P));
    P.n_envs = n_envs; P.nb = nb; P.adim = adim; P.chest = -1; P.wave_budget = wave_budget; P.has_obj = nb > 0; P.near_r = 0.065f; P.fd_div = PMG_FD_DIV;
    const float lo[3] = {-0.67f, -0.2f, 0.175f}, hi[3] = {-0.37f, 0.2f, 0.55f};
    for (int a = 0; a < 3; a++) { P.ee_lo[a] = lo[a]; P.ee_hi[a] = hi[a]; }
    P.hot = const_cast<float*>(hot);
    P.blocks = const_cast<float*>(blocks);
    const int nwg = (n_envs + PLAN_THREADS - 1) / PLAN_THREADS;
    std::vector<int> sc(4 + 3 * (size_t)n_envs + 3 * (size_t)nwg, -1);
    P.sched = sc.data();
    if (!two_pass) {
        if (n_envs > PLAN_MAX_TILES * 64) return -1;   /* (the launcher switches to two passes at PLAN_SINGLE_MAX already) */
        emu::launch(1, PLAN_THREADS, [&]() { plan_all(P, actions); });
    } else {
        emu::launch(nwg, PLAN_THREADS, [&]() { plan_count(P, actions); });
        emu::launch(nwg, PLAN_THREADS, [&]() { plan_scatter(P, actions); });
    }
    memcpy(sched_out, sc.data(), sizeof(int) * (3 + 3 * (size_t)n_envs));
    return *plan_promoted(P);                          /* 0 / 1: the word pmg_k_step_list reads for its issue priority */
}
