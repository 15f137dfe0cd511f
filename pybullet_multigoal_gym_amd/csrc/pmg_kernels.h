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

namespace pmg {

#ifdef PMG_PROFILE
#define PMG_TICK(i) do { long long t_ = wall_clock64(); if (wv::lane() == 0 && blockIdx.x == 0) P.prof[i] += t_ - tprev; tprev = wall_clock64(); } while (0)
#else
#define PMG_TICK(i) do { } while (0)
#endif

struct EnvParams {
    int n_envs, task, nb, grasping, has_obj, joint_control, binary_reward, max_steps, in_air, random_order;
    int multi;              /* multi-block observation layout: block_stack / block_rearrange */
    int curriculum, curriculum_update; /* kuka_multi_step_base_env.py:121-152 */
    int decomposition, grip_goal;      /* task_decomposition, grip_informed_goal (goal = blocks | tip target | finger width) */
    double goals_per_curriculum;
    int adim, odim, pdim, gdim, packed;
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
    int* sched;      /* [3 + 3N]: counts of {contact-prone, other} envs, the two env lists, then the redo count + list
                        of the row-packed path (pmg_packed.h) */
#ifdef PMG_PROFILE
    long long* prof; /* per-phase wall_clock64 ticks of env 0 */
#endif
};

/* ------------------------------------------------------------------ */
/* component i of the desired goal.  The multi-block tasks re-derive it from the current block poses at
 * every observation (kuka_multi_step_base_env.py:309-312): a block beyond the active curriculum level /
 * sub-goal index "is already at its goal".  cold[7] = level, goal[15] = moved-block mask.          */
constexpr int CURR_DIM = 16;
__device__ __forceinline__ float effective_goal_at_level(const EnvParams& P, int env, int i, int level)
{
    const float* g = P.goal + (size_t)env * GOAL_DIM;
    if (!P.multi) return g[i];
    const float* cold = P.cold + (size_t)env * COLD_DIM;
    const float* bb = P.blocks + (size_t)env * BLOCK_DIM * P.nb;
    int b = i / 3, a = i - 3 * b;
    if (P.task == PMG_TASK_BLOCK_STACK) {
        /* plain / curriculum: the first level+1 blocks of the order sit at their targets.  grip-informed sub-goals
         * come in (pick, place) pairs per block j (kuka_multi_step_envs.py:91-111): pick keeps blocks i < j, place
         * blocks i <= j at their targets; the tail is the gripper tip target and the finger width 0.03 */
        const bool pairs = P.grip_goal && P.decomposition;
        const int j = pairs ? level / 2 : level;
        const bool pick = pairs && (level % 2 == 0);
        if (i >= 3 * P.nb) {
            if (i == 3 * P.nb + 3) return 0.03f;
            int bj = (int)cold[8 + j];
            return pick ? bb[BLOCK_DIM * bj + a] : g[3 * bj + a];
        }
        int pos = 0;
        for (int s = 0; s < P.nb; s++) pos = ((int)cold[8 + s] == b) ? s : pos;   /* place of block b in the stack order */
        return (pick ? pos < j : pos <= j) ? g[i] : bb[BLOCK_DIM * b + a];
    }
    int moved = (int)g[15];
    if (!((moved >> b) & 1)) return bb[BLOCK_DIM * b + a];
    int kth = __popc((unsigned)moved & ((1u << b) - 1u));                        /* the k-th moved block takes target k */
    return g[3 * kth + a];
}
__device__ __forceinline__ float effective_goal(const EnvParams& P, int env, int i)
{
    return effective_goal_at_level(P, env, i, P.multi ? (int)P.cold[(size_t)env * COLD_DIM + 7] : 0);
}

/* ------------------------------------------------------------------ */
/* observation / reward pack                                           */
/* single-step tasks: kuka_single_step_base_env.py:193-244; robot state: kuka.py:227-256 */
__device__ __forceinline__ void write_outputs(const EnvParams& P, int env, const LaneConst& c, float q, float qd,
                                              int elapsed, bool with_reward)
{
    int l = wv::lane();
    Kin k;
    fk(c, q, k);
    float tip[3], Rt[9];
    tip_frame(k, tip, Rt);
    /* spatial velocity of every link; lane 6 = link 7 (tip, gripper base), lane 7 = finger 1 */
    float v[6];
#pragma unroll
    for (int a = 0; a < 6; a++) v[a] = chain_prefix(k.S[a] * qd);
    float v6[6];
    wv::bcastn<6>(v, 6, v6);
    float tv[3], t[3];
    cross3(v6, tip, t);
    tv[0] = v6[3] + t[0]; tv[1] = v6[4] + t[1]; tv[2] = v6[5] + t[2];
    float closeness = 0.f, fvel = 0.f;
    if (P.grasping) {
        /* tabs: finger frame origin -/+ 0.005 along finger y (urdf:480-494) */
        float tab[3];
        float sgn = l == 7 ? -0.005f : 0.005f;
#pragma unroll
        for (int a = 0; a < 3; a++) tab[a] = k.p[a] + k.R[3 * a + 1] * sgn;
        float vt[3];
        cross3(v, tab, vt);
        float pack[4] = {tab[0], tab[1], tab[2], v[4] + vt[1]}, t1[4], t2[4];
        wv::bcastn<4>(pack, 7, t1);
        wv::bcastn<4>(pack, 8, t2);
        float d[3] = {t1[0] - t2[0], t1[1] - t2[1], t1[2] - t2[2]};
        closeness = sqrtf(dot3(d, d));
        /* gripper base origin: link 7 + 0.055 z (urdf:390-395) */
        float gb[3] = {tip[0] + Rt[2] * (0.055f - TIP_Z), tip[1] + Rt[5] * (0.055f - TIP_Z), tip[2] + Rt[8] * (0.055f - TIP_Z)};
        float vb[3];
        cross3(v6, gb, vb);
        fvel = (v6[4] + vb[1]) - t1[3];
    }
    float* o = P.out + (size_t)env * P.packed;
    const float* g = P.goal + (size_t)env * GOAL_DIM;
    int jo = P.joint_control ? 7 : 0;
    float* obs = o;
    float* pol = o + P.odim;
    float* ag = pol + P.pdim;
    float* dg = ag + P.gdim;
    float* tail = dg + P.gdim;
    if (jo && l < 7) { obs[l] = q; pol[l] = q; }
    float agv[3] = {tip[0], tip[1], tip[2]};
    if (P.task == PMG_TASK_REACH) {
        if (l < 3) { float x = l == 0 ? tip[0] : (l == 1 ? tip[1] : tip[2]); obs[jo + l] = x; pol[jo + l] = x; ag[l] = x; }
    } else if (!P.multi) {
        const float* b = P.blocks + (size_t)env * BLOCK_DIM * P.nb;
        float bp[3] = {b[0], b[1], b[2]}, bv[3] = {b[7], b[8], b[9]}, bw[3] = {b[10], b[11], b[12]};
        if (l == 0) {
            float* s = obs + jo;
            s[0] = tip[0]; s[1] = tip[1]; s[2] = tip[2];
            s[3] = bp[0]; s[4] = bp[1]; s[5] = bp[2];
            s[6] = closeness;
            s[7] = tip[0] - bp[0]; s[8] = tip[1] - bp[1]; s[9] = tip[2] - bp[2];
            s[10] = tv[0]; s[11] = tv[1]; s[12] = tv[2];
            s[13] = fvel;
            s[14] = tv[0] - bv[0]; s[15] = tv[1] - bv[1]; s[16] = tv[2] - bv[2];
            s[17] = v6[0] - bw[0]; s[18] = v6[1] - bw[1]; s[19] = v6[2] - bw[2];
            float* ps = pol + jo;
            ps[0] = tip[0]; ps[1] = tip[1]; ps[2] = tip[2]; ps[3] = closeness;
            ps[4] = tip[0] - bp[0]; ps[5] = tip[1] - bp[1]; ps[6] = tip[2] - bp[2];
            ag[0] = bp[0]; ag[1] = bp[1]; ag[2] = bp[2];
        }
        agv[0] = bp[0]; agv[1] = bp[1]; agv[2] = bp[2];
    } else {
        /* block stack: kuka_multi_step_base_env.py:255-336, clipped to +-5 */
        const float* bb = P.blocks + (size_t)env * BLOCK_DIM * P.nb;
        if (l == 0) {
            float* s = obs + jo;
            s[0] = tip[0]; s[1] = tip[1]; s[2] = tip[2]; s[3] = closeness;
            s[4] = tv[0]; s[5] = tv[1]; s[6] = tv[2]; s[7] = fvel;
            float* ps = pol + jo;
            ps[0] = tip[0]; ps[1] = tip[1]; ps[2] = tip[2]; ps[3] = closeness;
        }
        if (l < P.nb) {
            const float* b = bb + BLOCK_DIM * l;
            float* s = obs + jo + 8 + 16 * l;
            float* ps = pol + jo + 4 + 3 * l;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                s[a] = b[a];
                s[3 + a] = tip[a] - b[a];
                ps[a] = tip[a] - b[a];
                s[10 + a] = tv[a] - b[7 + a];
                s[13 + a] = v6[a] - b[10 + a];
                ag[3 * l + a] = b[a];
            }
#pragma unroll
            for (int a = 0; a < 4; a++) s[6 + a] = b[3 + a];
        }
        if (P.grip_goal && l == 0) { /* kuka_multi_step_base_env.py:300-304 */
            float* t = ag + 3 * P.nb;
            t[0] = tip[0]; t[1] = tip[1]; t[2] = tip[2]; t[3] = closeness;
        }
    }
    wv::lds_sync();
    if (P.multi) {
        for (int i = l; i < P.odim; i += 64) obs[i] = fminf(fmaxf(obs[i], -5.f), 5.f);
        for (int i = l; i < P.pdim; i += 64) pol[i] = fminf(fmaxf(pol[i], -5.f), 5.f);
    }
    float dgl = l < P.gdim ? effective_goal(P, env, l) : 0.f;
    if (l < P.gdim) dg[l] = dgl;
    if (with_reward) {
        /* _compute_reward: d = ||ag - dg||, binary -(d > thr) as float32 or dense -d */
        float dd = 0.f;
        if (P.multi) {
            const float* bb = P.blocks + (size_t)env * BLOCK_DIM * P.nb;
            float agl = 0.f;
            if (l < 3 * P.nb) agl = bb[BLOCK_DIM * (l / 3) + l % 3];
            else if (l < P.gdim) { int t = l - 3 * P.nb; agl = t == 0 ? tip[0] : (t == 1 ? tip[1] : (t == 2 ? tip[2] : closeness)); }
            float e = l < P.gdim ? agl - dgl : 0.f;
            dd = wv::sum_rows<2>(e * e);   /* goal_dim <= 19: lanes of the first two rows */
        } else {
#pragma unroll
            for (int a = 0; a < 3; a++) { float e = agv[a] - g[a]; dd += e * e; }
        }
        float d = sqrtf(dd);
        bool not_achieved = d > P.thr;
        if (l == 0) {
            tail[0] = P.binary_reward ? (not_achieved ? -1.f : -0.f) : -d;
            tail[1] = not_achieved ? 0.f : 1.f;
            tail[2] = elapsed >= P.max_steps ? 1.f : 0.f;
        }
    } else if (l == 0) {
        tail[0] = 0.f; tail[1] = 0.f; tail[2] = 0.f;
    }
}

/* ------------------------------------------------------------------ */
/* contact phases of a substep.  For the reach kernel (no free bodies) they
 * are rare (fingers at the table clip plane only), so they are kept out of
 * line there to keep the register budget of the 100-substep loop small. */
template <int NB, int MAXC>
__device__ __noinline__ int collide_cold(ContactLds<NB, MAXC>& L, int nb, float tcx, float tcy, float tcz, float thx, float thy,
                                         float thz, float tmu)
{
    float tc[3] = {tcx, tcy, tcz}, th[3] = {thx, thy, thz};
    return collide<NB, MAXC, false>(L, nb, tc, th, tmu);
}
/* publish (inline, from registers) what the pair / contact lanes need, then run the narrowphase */
template <int NB, int MAXC, bool CYL>
__device__ __forceinline__ int detect(ContactLds<NB, MAXC>& L, int nb, const Kin& k, float tcx, float tcy, float tcz, float thx,
                                      float thy, float thz, float tmu)
{
    int l = wv::lane();
    if (l == 7 || l == 8) {
        float* f = L.fing[l - 7];
#pragma unroll
        for (int a = 0; a < 3; a++) f[a] = k.p[a];
#pragma unroll
        for (int a = 0; a < 9; a++) f[3 + a] = k.R[a];
    }
    if (l < NJ) {
#pragma unroll
        for (int a = 0; a < 6; a++) L.S[l][a] = k.S[a];
    }
    if (NB > 0 && l == 6) { /* gripper-base cylinder: link 7 frame shifted 0.055 along its z (urdf:390-395) */
#pragma unroll
        for (int a = 0; a < 3; a++) L.gbase[a] = k.p[a] + k.R[3 * a + 2] * 0.055f;
#pragma unroll
        for (int a = 0; a < 9; a++) L.gbase[3 + a] = k.R[a];
    }
    if (l < nb) quat_to_R(L.blk[l] + 3, L.blkR[l]);
    wv::lds_sync();
    if (PMG_COLD_CONTACTS && NB == 0) return collide_cold<NB, MAXC>(L, nb, tcx, tcy, tcz, thx, thy, thz, tmu);
    float tc[3] = {tcx, tcy, tcz}, th[3] = {thx, thy, thz};
    return collide<NB, MAXC, CYL>(L, nb, tc, th, tmu);
}

template <int NB, int MAXC>
__device__ __noinline__ void build_rows_cold(ContactLds<NB, MAXC>& L, int nc)
{
    build_contact_rows<NB, MAXC, false>(L, nc);
}
template <int NB, int MAXC, bool CYL>
__device__ __forceinline__ void prepare_rows(ContactLds<NB, MAXC>& L, const float* minv, float qd, int nc)
{
    int l = wv::lane();
    if (l < NJ) {
#pragma unroll
        for (int j = 0; j < NJ; j++) L.minv[l][j] = minv[j];
        L.qd[l] = qd;
    }
    wv::lds_sync();
    if (PMG_COLD_CONTACTS && NB == 0) build_rows_cold<NB, MAXC>(L, nc);
    else build_contact_rows<NB, MAXC, CYL>(L, nc);
}

/* reach: PGS iterations when finger x table contacts exist -- register-resident rows (RobotRows) */
template <int NB, int MAXC>
__device__ __forceinline__ void reach_contact_pgs(ContactLds<NB, MAXC>& L, int nc, NcRows& r, const float* minv_in, float& dv)
{
    float minv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) minv[j] = minv_in[j];
    RobotRows<MAXC> rr;
    load_robot_rows(L, nc, rr);
    for (int it = 0; it < SOLVER_ITERS; it++) {
        nc_sweep(r, (it & 1) != 0, minv, dv);
        float resid = robot_rows_iteration(rr, nc, dv);
        resid = fmaxf(resid, wv::max_row0(nc_residual(r)));
        if (resid <= RESIDUAL_THRESHOLD) break;
    }
}

/* ------------------------------------------------------------------ */
/* one 2 ms substep: collide, unconstrained velocities, PGS rows, integrate
 * ([BULLET-PRIOR] btMultiBodyDynamicsWorld::internalSingleStepSimulation)  */
template <int NB, int MAXC, bool CYL>
__device__ __forceinline__ void substep(const EnvParams& P, ContactLds<NB, MAXC>& L, const LaneConst& c_in, float& q, float& qd,
                                        float tau, float mtarget, float mimp)
{
    int l = wv::lane();
    const int nb = NB > 0 ? P.nb : 0;
#ifdef PMG_PROFILE
    long long tprev = wall_clock64();
#endif
    LaneConst c = c_in;
    wv::opaque(c.col); /* keep the LDS constant reads inside the loop (no 40-register hoist) */
    Kin k;
    fk(c, q, k);
    PMG_TICK(0);
    /* contact detection; without free bodies it is skipped (wave-uniform) unless a finger is near the table */
    int nc = 0;
    bool low = (l == 7 || l == 8) && (finger_zmin(k.p, k.R) < P.table_c[2] + P.table_h[2] + CONTACT_MARGIN);
    if (NB > 0 || wv::ballot(low) != 0ull)
        nc = detect<NB, MAXC, CYL>(L, nb, k, P.table_c[0], P.table_c[1], P.table_c[2], P.table_h[0], P.table_h[1], P.table_h[2], P.table_mu);

    PMG_TICK(1);
    /* unconstrained velocity update: robot (CRBA + RNEA) ... */
    float I10[10], minv[NJ];
    body_inertia(c, k, I10);
    if (l >= NJ) {
#pragma unroll
        for (int a = 0; a < 10; a++) I10[a] = 0.f;
    }
    float h = bias_torque(c, k, I10, qd);   /* bias first: its temporaries are dead before the matrix work */
    mass_inverse(k, I10, minv);
    float rq = l < NJ ? tau - h : 0.f;
    float qdd = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) qdd += minv[j] * wv::bcast(rq, j);
    qd += DT * qdd;
    /* ... and the free objects: gravity, Bullet's base damping and the gyroscopic term w x (I w)
     * (zero for the isotropic cubes, not for the slide puck) */
    if (l < nb) {
        float* b = L.blk[l];
        const float* Rm = L.blkR[l];
        float kl = LINK_DAMPING * (1.f + sqrtf(dot3(b + 7, b + 7))), ka = LINK_DAMPING * (1.f + sqrtf(dot3(b + 10, b + 10)));
        float al[3] = {-b[10] * ka, -b[11] * ka, -b[12] * ka};
        if (CYL) {
            float wl[3], Iw[3], gy[3], tq[3];
            wl[0] = Rm[0] * b[10] + Rm[3] * b[11] + Rm[6] * b[12];
            wl[1] = Rm[1] * b[10] + Rm[4] * b[11] + Rm[7] * b[12];
            wl[2] = Rm[2] * b[10] + Rm[5] * b[11] + Rm[8] * b[12];
#pragma unroll
            for (int a = 0; a < 3; a++) Iw[a] = wl[a] / ObjT<CYL>::inv_inertia(a);
            cross3(wl, Iw, gy);
#pragma unroll
            for (int a = 0; a < 3; a++) tq[a] = (-Iw[a] * ka - gy[a]) * ObjT<CYL>::inv_inertia(a);
            mat3v(Rm, tq, al);
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            b[7 + a] += DT * (-b[7 + a] * kl + (a == 2 ? -GRAVITY : 0.f));
            b[10 + a] += DT * al[a];
        }
    }
    PMG_TICK(2);
    if (nc > 0) prepare_rows<NB, MAXC, CYL>(L, minv, qd, nc);
    PMG_TICK(3);
    NcRows r;
    build_nc_rows(c, minv, q, qd, mtarget, mimp, r);
    float dv = 0.f; /* lanes 0..8: joint velocity change; lanes 9+6b+c: component c of block b */
    if (NB == 0) {
        if (nc > 0) {
            reach_contact_pgs<NB, MAXC>(L, nc, r, minv, dv);
        } else {
            for (int it = 0; it < SOLVER_ITERS; it++) {
                nc_sweep(r, (it & 1) != 0, minv, dv);
                if (wv::max_row0(nc_residual(r)) <= RESIDUAL_THRESHOLD) break;
            }
        }
    } else {
        LaneDof<NB> ld;
        lane_dof(ld);
        for (int it = 0; it < SOLVER_ITERS; it++) {
            PMG_TICK(10);
            nc_sweep(r, (it & 1) != 0, minv, dv);
            PMG_TICK(8);
            float resid = nc > 0 ? lds_rows_iteration(L, nc, ld, dv) : 0.f;
            PMG_TICK(9);
            resid = fmaxf(resid, nc_residual(r));
            /* residuals are per lane (valid where the DoFs live): one reduction per iteration */
            if (wv::max_all(resid) <= RESIDUAL_THRESHOLD) break;
        }
    }
    PMG_TICK(4);
    if (l < NJ) qd += dv;
    q += DT * qd;
    if (NB > 0) {
        if (l >= NJ && l < NJ + 6 * nb) L.blk[(l - NJ) / 6][7 + (l - NJ) % 6] += dv;
        wv::lds_sync();
        if (l < nb) {
            float* b = L.blk[l];
#pragma unroll
            for (int a = 0; a < 3; a++) b[a] += DT * b[7 + a];
            /* quat <- exp(omega dt) * quat */
            float wn = sqrtf(dot3(b + 10, b + 10));
            float ang = wn * DT;
            float dq[4] = {0.f, 0.f, 0.f, 1.f};
            if (ang > 1e-12f) {
                float sh, ch;
                sincosf(0.5f * ang, &sh, &ch);
                float s = sh / wn;
                dq[0] = b[10] * s; dq[1] = b[11] * s; dq[2] = b[12] * s; dq[3] = ch;
            }
            float x = dq[3] * b[3] + dq[0] * b[6] + dq[1] * b[5] - dq[2] * b[4];
            float y = dq[3] * b[4] + dq[1] * b[6] + dq[2] * b[3] - dq[0] * b[5];
            float z = dq[3] * b[5] + dq[2] * b[6] + dq[0] * b[4] - dq[1] * b[3];
            float w = dq[3] * b[6] - dq[0] * b[3] - dq[1] * b[4] - dq[2] * b[5];
            float inv = 1.f / sqrtf(x * x + y * y + z * z + w * w);
            b[3] = x * inv; b[4] = y * inv; b[5] = z * inv; b[6] = w * inv;
        }
        wv::lds_sync();
    }
}

/* ------------------------------------------------------------------ */
/* Launch-order plan.  A batched step lasts as long as its slowest wavefront, and the slow ones are
 * the envs whose fingers will touch the table (contact phases, ~2.5x a contact-free substep).  The
 * hardware favours the OLDEST wave of a SIMD, so those envs are given the lowest workgroup ids:
 * dispatched first, one per SIMD, they run at single-wave speed from t = 0 while the rest fill the
 * issue slots.  The mapping never changes a result (envs are independent), only who waits.     */
__device__ __forceinline__ bool contact_prone(const EnvParams& P, const float* actions, int env)
{
    if (P.nb > 0 || P.joint_control) return true;         /* blocks always touch the table */
    const float* hot = P.hot + (size_t)env * HOT_DIM;
    float z = hot[20];                                    /* tip target: the tip is within mm of it */
    float zn = fminf(fmaxf(z + actions[(size_t)env * P.adim + 2] * 0.01f, P.ee_lo[2]), P.ee_hi[2]);
    return fminf(z, zn) < P.ee_lo[2] + 0.012f;
}
/* one 1024-thread workgroup partitions all envs (stable, no atomics): per-wave ballots, counts of
 * every (chunk, wave) tile in LDS, then each tile scatters at its exclusive prefix */
constexpr int PLAN_THREADS = 1024, PLAN_MAX_TILES = 1024;
__device__ __forceinline__ void plan_all(const EnvParams& P, const float* actions)
{
    __shared__ int cnt0[PLAN_MAX_TILES], cnt1[PLAN_MAX_TILES];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = PLAN_THREADS / 64;
    const int chunks = (P.n_envs + PLAN_THREADS - 1) / PLAN_THREADS;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c = 0; c < chunks; c++) {
        int env = c * PLAN_THREADS + tid;
        bool valid = env < P.n_envs;
        bool prone = valid && contact_prone(P, actions, env);
        unsigned long long m0 = wv::ballot(prone), m1 = wv::ballot(valid && !prone);
        if (lane == 0 && c * waves + wave < PLAN_MAX_TILES) { cnt0[c * waves + wave] = __popcll(m0); cnt1[c * waves + wave] = __popcll(m1); }
    }
    __syncthreads();
    const int tiles = chunks * waves < PLAN_MAX_TILES ? chunks * waves : PLAN_MAX_TILES;
    for (int c = 0; c < chunks; c++) {
        int tile = c * waves + wave;
        int env = c * PLAN_THREADS + tid;
        bool valid = env < P.n_envs && tile < PLAN_MAX_TILES;
        bool prone = valid && contact_prone(P, actions, env);
        unsigned long long m0 = wv::ballot(prone), m1 = wv::ballot(valid && !prone);
        int b0 = 0, b1 = 0;
        for (int t = 0; t < tile && t < tiles; t++) { b0 += cnt0[t]; b1 += cnt1[t]; }
        if (prone) P.sched[2 + b0 + __popcll(m0 & below)] = env;
        else if (valid) P.sched[2 + P.n_envs + b1 + __popcll(m1 & below)] = env;
    }
    if (tid == 0) {
        int n0 = 0, n1 = 0;
        for (int t = 0; t < tiles; t++) { n0 += cnt0[t]; n1 += cnt1[t]; }
        P.sched[0] = n0;
        P.sched[1] = n1;
        P.sched[2 + 2 * P.n_envs] = 0; /* redo list of the row-packed path starts empty */
    }
}
__device__ __forceinline__ int scheduled_env(const EnvParams& P, int block)
{
    int n0 = P.sched[0];
    return block < n0 ? P.sched[2 + block] : P.sched[2 + P.n_envs + (block - n0)];
}

/* ------------------------------------------------------------------ */
/* env.step(): kuka.py:167-225 + _get_obs + _compute_reward + TimeLimit */
template <int NB, int MAXC, bool CYL>
__device__ __forceinline__ void step_env(const EnvParams& P, const float* actions, int env)
{
    __shared__ ContactLds<NB, MAXC> L;
    __shared__ LaneTabStore lcs;
    int l = wv::lane();
    if (env >= P.n_envs) return;
    LaneConst c;
    load_lane_const(lcs, c);
    float* hot = P.hot + (size_t)env * HOT_DIM;
    int ll = l < NJ ? l : 0;
    float q = hot[ll], qd = hot[9 + ll];
    if (l >= NJ) { q = 0.f; qd = 0.f; }
    const int nb = NB > 0 ? P.nb : 0;
    float* gblk = P.blocks + (size_t)env * BLOCK_DIM * P.nb;
    for (int i = l; i < BLOCK_DIM * nb; i += 64) L.blk[i / BLOCK_DIM][i % BLOCK_DIM] = gblk[i];
#ifdef PMG_PROFILE
    long long tprev = wall_clock64();
#endif
    const float* act = actions + (size_t)env * P.adim;
    float grip = hot[28];
    int elapsed = (int)hot[29];
    if (P.grasping) grip = (float)(((double)act[P.adim - 1] + 1.0) * (0.035 / 2)); /* kuka.py:171 */
    float mtarget = grip, mimp = FINGER_FORCE * PHYSICS_DT;
    float ee[3] = {hot[18], hot[19], hot[20]};
    float jt = l < 7 ? hot[21 + l] : 0.f;
    if (P.joint_control) {
        if (l < 7) jt = act[l] * 0.05f + jt; /* kuka.py:205 */
        if (l < 7) mtarget = jt;
    } else {
#pragma unroll
        for (int a = 0; a < 3; a++) {      /* kuka.py:209-212 */
            float t = ee[a] + act[a] * 0.01f;
            ee[a] = fminf(fmaxf(t, P.ee_lo[a]), P.ee_hi[a]);
        }
        float qik = ik_solve(c, q, ee);    /* kuka.py:214 */
        if (l < 7) mtarget = qik;
    }
    if (l < 7) mimp = ARM_FORCE * PHYSICS_DT; /* kuka.py:282-290 */
    wv::lds_sync();
    PMG_TICK(5);
    for (int s = 0; s < SIM_STEPS; s++) {  /* kuka.py:223-225 */
        float tau = -c.jdamp() * qd;         /* joint damping latched per stepSimulation */
        for (int ss = 0; ss < SUBSTEPS; ss++) substep<NB, MAXC, CYL>(P, L, c, q, qd, tau, mtarget, mimp);
    }
    PMG_TICK(6);
    elapsed++;
    if (l < NJ) { hot[l] = q; hot[9 + l] = qd; }
    if (l < 3) hot[18 + l] = l == 0 ? ee[0] : (l == 1 ? ee[1] : ee[2]);
    if (l < 7) hot[21 + l] = jt;
    if (l == 0) { hot[28] = grip; hot[29] = (float)elapsed; hot[30] = 1.f; }
    for (int i = l; i < BLOCK_DIM * nb; i += 64) gblk[i] = L.blk[i / BLOCK_DIM][i % BLOCK_DIM];
    wv::lds_sync();
    write_outputs(P, env, c, q, qd, elapsed, true);
    PMG_TICK(7);
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
        if (P.task == PMG_TASK_BLOCK_STACK) {
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
            float* cs = P.curr + (size_t)env * CURR_DIM;
            double cdf[5], acc = 0.0;
            for (int i = 0; i < P.nb; i++) { acc += (double)cs[i]; cdf[i] = acc; }
            double u = mt_uniform(mt, 0.0, 1.0);
            level = 0;
            while (level < P.nb && cdf[level] / acc <= u) level++;
            if (level > P.nb - 1) level = P.nb - 1;
            cs[10] = (float)(level * 25 + 50);
            if (P.task == PMG_TASK_BLOCK_REARRANGE) { /* choice(arange(nb), size=level+1, replace=False) = permutation(nb)[:level+1] */
                int perm[5] = {0, 1, 2, 3, 4};
                for (int i = P.nb - 1; i >= 1; i--) {
                    unsigned j = mt_interval(mt, (unsigned)i);
                    int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
                }
                moved = 0;
                for (int i = 0; i <= level; i++) moved |= 1 << perm[i];
            }
            if (P.curriculum_update) { /* _update_curriculum_prob: kuka_multi_step_base_env.py:350-379 */
                cs[5 + level] += 1.f;
                bool fin[5], half[5];
                for (int i = 0; i < P.nb; i++) {
                    fin[i] = (double)cs[5 + i] >= P.goals_per_curriculum;
                    half[i] = (double)cs[5 + i] >= P.goals_per_curriculum / 2;
                    if (fin[i]) cs[i] = 0.f;
                }
                if (half[0] && !fin[0]) { cs[0] = 0.5f; cs[1] = 0.5f; }
                for (int i = 1; i < P.nb - 1; i++)
                    if (fin[i - 1] && !fin[i]) {
                        if (half[i]) { cs[i] = 0.5f; cs[i + 1] = 0.5f; }
                        else cs[i] = 1.f;
                    }
                if (fin[P.nb - 2]) cs[P.nb - 1] = 1.f;
            }
        }
        for (int b = 0; b < 5; b++) cold[8 + b] = (float)order[b];
        cold[7] = (float)level;
        g[15] = (float)moved;
    }
}

/* env.reset(): kuka.py:120-165 + the task reset above + _get_obs */
__device__ __forceinline__ void reset_env(const EnvParams& P, const unsigned char* mask)
{
    __shared__ LaneTabStore lcs;
    int env = (int)blockIdx.x, l = wv::lane();
    if (env >= P.n_envs) return;
    LaneConst c;
    load_lane_const(lcs, c);
    float* hot = P.hot + (size_t)env * HOT_DIM;
    float* cold = P.cold + (size_t)env * COLD_DIM;
    int ll = l < NJ ? l : 0;
    bool doit = mask == nullptr || mask[env] != 0;
    float q, qd;
    int elapsed;
    if (doit) {
        q = l < 7 ? cold[l] : (l < NJ ? FINGER_LIMIT : 0.f); /* kuka.py:158,161 */
        qd = 0.f;
        float tgt[3] = {(float)P.tip_init[0], (float)P.tip_init[1], (float)P.tip_init[2]};
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
    write_outputs(P, env, c, q, qd, elapsed, false);
}

}  // namespace pmg
#include "pmg_packed.h"
#endif
