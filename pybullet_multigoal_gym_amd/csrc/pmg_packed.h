/*
 * pmg_packed.h -- the contact-free reach step with FOUR environments per wavefront.
 *
 * The robot maths of pmg_device_body.inc keeps one link / DoF per lane and needs 9 of them; its scans and
 * butterflies are DPP operations inside a 16-lane row.  A wavefront therefore carries four independent
 * environments, one per row (namespace pmgp = the same source compiled against the row-local primitives
 * wr), which divides the VALU instruction count per env-step -- the resource that binds the reach kernel
 * -- by four.
 *
 * Only envs the launch-order plan classified as away from the table take this path.  The classification is
 * a prediction, so every substep still evaluates the exact "finger within the contact margin of the table"
 * predicate of the one-env-per-wave kernel; a row that trips it stops, writes NOTHING and queues its env on
 * the redo list, and pmg_k_redo recomputes that env from its untouched state with the full contact path.
 * Results are therefore the unpacked kernel's (to fp32 rounding), whatever the prediction said.
 */
#ifndef PMG_PACKED_H
#define PMG_PACKED_H

namespace pmgp {

#ifdef PMG_PROFILE
#define PMGP_T0() long long pt_ = prof::now()
#define PMGP_T(i) PMG_STAMP(pt_, i)
#else
#define PMGP_T0() do { } while (0)
#define PMGP_T(i) do { } while (0)
#endif

/* one 2 ms substep of a contact-free env; false = a finger reached the contact margin (caller must redo) */
__device__ __forceinline__ bool substep_free(const EnvParams& P, const LaneConst& c_in, float& q, float& qd, float tau,
                                             float mtarget, float mimp)
{
    const int l = wr::lane();
    LaneConst c = c_in;
    wr::opaque(c.col); /* keep the LDS constant reads inside the loop (no 40-register hoist) */
    PMGP_T0();
    Kin k;
    fk(c, q, k);
    PMGP_T(0);
    bool low = (l == 7 || l == 8) && (pmg::finger_zmin(k.p, k.R) < P.table_c[2] + P.table_h[2] + pmg::CONTACT_MARGIN);
    if (wr::ballot(low) != 0ull) return false;
    float I10[10], minv[NJ];
    body_inertia(c, k, I10);
    if (l >= NJ) {
#pragma unroll
        for (int a = 0; a < 10; a++) I10[a] = 0.f;
    }
    PMGP_T(1);
    float h = bias_torque(c, k, I10, qd);
    PMGP_T(2);
    mass_inverse(k, I10, minv);
    PMGP_T(3);
    float rq = l < NJ ? tau - h : 0.f;
    float qdd = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; j++) qdd += minv[j] * wr::bcast_r0(rq, j);
    qd += DT * qdd;
    NcRows r;
    build_nc_rows(c, minv, q, qd, mtarget, mimp, r);
    PMGP_T(4);
    float dv = 0.f;
    McState mc;
    mc_init(r, minv, mc);
    nc_solve_free(r, mc, minv, dv);
    PMGP_T(5);
    if (l < NJ) qd += dv;
    q += DT * qd;
    return true;
}

/* observation / reward pack of the reach task (kuka_single_step_base_env.py:193-244), one env per row */
__device__ __forceinline__ void write_outputs_reach(const EnvParams& P, int env, const LaneConst& c, float q, int elapsed)
{
    const int l = wr::lane();
    Kin k;
    fk(c, q, k);
    float tip[3], Rt[9];
    tip_frame(k, tip, Rt);
    float* o = P.out + (size_t)env * P.packed;
    const float* g = P.goal + (size_t)env * GOAL_DIM;
    float* obs = o;
    float* pol = o + P.odim;
    float* ag = pol + P.pdim;
    float* dg = ag + P.gdim;
    float* tail = dg + P.gdim;
    const int jo = P.joint_control ? 7 : 0;
    if (jo && l < 7) { obs[l] = q; pol[l] = q; }                /* joint angles first: kuka_single_step_base_env.py:214-216 */
    if (l < 3) {
        float x = l == 0 ? tip[0] : (l == 1 ? tip[1] : tip[2]);
        obs[jo + l] = x; pol[jo + l] = x; ag[l] = x;
        dg[l] = g[l];
    }
    float dd = 0.f;
#pragma unroll
    for (int a = 0; a < 3; a++) { float e = tip[a] - g[a]; dd += e * e; }
    float d = sqrtf(dd);
    bool not_achieved = d > P.thr;
    if (l == 0) {
        tail[0] = P.binary_reward ? (not_achieved ? -1.f : -0.f) : -d;
        tail[1] = not_achieved ? 0.f : 1.f;
        tail[2] = elapsed >= P.max_steps ? 1.f : 0.f;
    }
}

/* env.step() for up to four envs of the contact-free list: kuka.py:167-225 + _get_obs + _compute_reward + TimeLimit */
__device__ __forceinline__ void step_group(const EnvParams& P, const float* actions, int group)
{
    __shared__ LaneTabStore lcs;
    const int l = wr::lane();
    const int n1 = P.sched[1];
    if (4 * group >= n1) return;                       /* wave-uniform: nothing queued for this wavefront */
    const int idx = 4 * group + wr::row();
    const bool have = idx < n1;                        /* surplus rows of the last wave shadow its last env and write nothing */
    const int env = P.sched[2 + P.n_envs + (have ? idx : n1 - 1)];
#ifdef PMG_PROFILE
    prof::begin();
#endif
    LaneConst c;
    load_lane_const(lcs, c);
    float* hot = P.hot + (size_t)env * HOT_DIM;
    const int ll = l < NJ ? l : 0;
    float q = hot[ll], qd = hot[9 + ll];
    if (l >= NJ) { q = 0.f; qd = 0.f; }
    const float* act = actions + (size_t)env * P.adim;
    const float grip = hot[28];
    int elapsed = (int)hot[29];
    float mtarget = grip, mimp = FINGER_FORCE * PHYSICS_DT;
    float ee[3] = {hot[18], hot[19], hot[20]};
    float jt = l < 7 ? hot[21 + l] : 0.f;
    PMGP_T0();
    if (P.joint_control) {
        if (l < 7) { jt = act[l] * 0.05f + jt; mtarget = jt; }   /* kuka.py:205 */
    } else {
#pragma unroll
        for (int a = 0; a < 3; a++) {                  /* kuka.py:209-212 */
            float t = ee[a] + act[a] * 0.01f;
            ee[a] = fminf(fmaxf(t, P.ee_lo[a]), P.ee_hi[a]);
        }
        float qik = ik_solve(c, q, ee);                /* kuka.py:214 */
        PMGP_T(7);
        if (l < 7) mtarget = qik;
    }
    if (l < 7) mimp = ARM_FORCE * PHYSICS_DT;          /* kuka.py:282-290 */
    bool ok = true;
    for (int s = 0; s < SIM_STEPS && ok; s++) {        /* kuka.py:223-225 */
        float tau = -c.jdamp() * qd;                   /* joint damping latched per stepSimulation */
        for (int ss = 0; ss < SUBSTEPS && ok; ss++) ok = substep_free(P, c, q, qd, tau, mtarget, mimp);
    }
    PMGP_T(8);
#ifdef PMG_PROFILE
    prof::flush(P.prof);
#endif
    if (!have) return;
    if (!ok) {                                         /* mispredicted: leave the state untouched, queue the env for pmg_k_redo */
        if (l == 0) {
            int* redo = P.sched + 2 + 2 * P.n_envs;
            int slot = atomicAdd(redo, 1);
            redo[1 + slot] = env;
        }
        return;
    }
    elapsed++;
    if (l < NJ) { hot[l] = q; hot[9 + l] = qd; }
    if (l < 3) hot[18 + l] = l == 0 ? ee[0] : (l == 1 ? ee[1] : ee[2]);
    if (l < 7) hot[21 + l] = jt;
    if (l == 0) { hot[29] = (float)elapsed; hot[30] = 1.f; }
    write_outputs_reach(P, env, c, q, elapsed);
}

/* ---------------------------------------------------------------- */
/* One free object (push / pick_and_place / slide), four envs per wavefront.  These kernels are bound by the
 * LATENCY of the serial constraint solve, not by issue slots, and a one-env wavefront needs two dispatch rounds
 * for 4096 envs; four envs per wavefront fit the batch in one round.  Each row gets its own contact store; to keep
 * four of them (plus the lane table) under 40 KB -- four workgroups per CU -- a row holds PACKED_MAXC contacts
 * instead of 24.  The launch-order plan keeps envs whose gripper works on the object (tip target within 8 cm of it:
 * the only situation with more contacts) on the one-env-per-wavefront list of the same fused kernel; a substep
 * that still finds more gives the env up exactly like a mispredicted reach env: nothing is written and
 * pmg_k_redo_obj recomputes it with the full kernel. */
constexpr int PACKED_MAXC = 12;
struct ObjLds4 { /* LDS of a packed workgroup: four contact stores + the lane-constant table */
    ContactLds<1, PACKED_MAXC> Ls[4];
    LaneTabStore lcs;
};
template <bool CYL>
__device__ __forceinline__ void step_group_obj(const EnvParams& P, const float* actions, int group, ObjLds4& sm)
{
    ContactLds<1, PACKED_MAXC>* Ls = sm.Ls;
    LaneTabStore& lcs = sm.lcs;
    const int n1 = P.sched[1];
    if (4 * group >= n1) return;
    const int idx = 4 * group + wr::row();
    const bool have = idx < n1;                        /* surplus rows shadow the last env and write nothing */
    const int env = P.sched[2 + P.n_envs + (have ? idx : n1 - 1)];
    const bool ok = step_env_core<1, PACKED_MAXC, CYL>(P, actions, env, Ls[wr::row()], lcs, have);
    if (have && !ok && wr::lane() == 0) {
        int* redo = P.sched + 2 + 2 * P.n_envs;
        int slot = atomicAdd(redo, 1);
        redo[1 + slot] = env;
    }
}

}  // namespace pmgp
#endif
