/*
 * pmg_device.h -- device code of the per-step hot path, one wavefront per env.
 *
 * Replaces, for N envs at once, what the reference does through PyBullet in
 * Kuka.apply_action (P/robots/kuka.py:167-225): tip-target integration and
 * clipping, calculateInverseKinematics, POSITION_CONTROL motors and
 * 5 x stepSimulation (= 100 substeps of 2 ms with 5 solver iterations,
 * P/envs/base_envs/base_env.py:203-220), followed by the state read-out of
 * Kuka.calc_robot_state (kuka.py:227-256).
 *
 * Layout: lane l < 9 owns movable link / DoF l (iiwa_joint_1..7, finger1,
 * finger2) and keeps its link frame, spatial axis, inertia, velocity and its
 * row of M^-1 in registers.  Chain recursions are DPP prefix/suffix scans in
 * the first 16-lane row, matrix work is lane = row with v_readlane
 * broadcasts; no MFMA (nothing here is a dense contraction), no HBM traffic
 * inside the 100-substep loop.
 *
 * Spatial algebra: world coordinates, Pluecker vectors referenced to the
 * world origin, motion = [w; v_O], force = [n_O; f].  A rigid-body inertia is
 * the 10-vector (m, H = m c, Ibar about the origin: xx xy xz yy yz zz), so the
 * composite-body recursion is a plain suffix SUM along the chain.
 * Dynamics = CRBA (mass matrix) + RNEA (bias) + in-register Gauss-Jordan
 * M^-1, mathematically identical to the articulated-body algorithm the
 * oracle restates from Bullet.
 */
#ifndef PMG_DEVICE_H
#define PMG_DEVICE_H

#include <pmg_wave.h> /* -I csrc (gfx950) or -I tests/emu (CPU emulator) */
#include "../../include/pmg.h"
#include "../../include/pmg_model.h"

namespace pmg {

constexpr int NJ = 9;
constexpr float DT = 0.002f;            /* base_env.py:217-219 */
constexpr int SUBSTEPS = 20;            /* base_env.py:219 */
constexpr int SIM_STEPS = 5;            /* kuka.py:223-225 */
constexpr int SOLVER_ITERS = 5;         /* base_env.py:37 */
constexpr float PHYSICS_DT = 0.04f;     /* base_env.py:217 */
constexpr float GRAVITY = 9.81f;        /* base_env.py:17 */
constexpr float CONTACT_ERP = 0.9f;     /* base_env.py:216 */
constexpr float JOINT_ERP = 0.2f;
constexpr float LINEAR_SLOP = 1e-5f;
constexpr float RESIDUAL_THRESHOLD = 1e-7f;
constexpr float LINK_DAMPING = 0.04f;
constexpr float LIMIT_MAX_IMPULSE = 100.f;
constexpr float ARM_KP = 0.03f, ARM_KD = 1.0f, ARM_FORCE = 200.f, FINGER_FORCE = 50.f; /* kuka.py:287-301 */
constexpr float FINGER_LIMIT = 0.035f;  /* kuka.py:71 */
constexpr int IK_MAX_ITER = 40;         /* kuka.py:278 */
constexpr float IK_THRESHOLD = 1e-5f;   /* kuka.py:279 */
constexpr float IK_DAMPING = 0.5f;
constexpr float IK_MAX_STEP = 0.78539816339744830962f;
constexpr float SIMD_EPS = 1.1920929e-07f;
constexpr float TIP_Z = 0.12f;          /* iiwa_gripper_tip_joint, urdf:311-315 */

/* state rows in HBM (float32, DESIGN.md "state rows") */
constexpr int HOT_DIM = 32;   /* q9 qd9 ee3 jt7 grip elapsed enabled resets */
constexpr int COLD_DIM = 16;  /* rest7 - order5 base3 */
constexpr int GOAL_DIM = 16;
constexpr int BLOCK_DIM = 13; /* pos3 quat4 vel3 omg3 */

__constant__ float C_JXYZ[NJ][3] = PMG_JXYZ;
__constant__ float C_JROT[NJ][3][3] = PMG_JROT;
__constant__ float C_JAXIS[NJ][3] = PMG_JAXIS;
__constant__ int C_JTYPE[NJ] = PMG_JTYPE;
__constant__ float C_JLO[NJ] = PMG_JLO;
__constant__ float C_JHI[NJ] = PMG_JHI;
__constant__ float C_JDAMP[NJ] = PMG_JDAMP;
__constant__ float C_MASS[NJ] = PMG_MB_MASS;
__constant__ float C_H[NJ][3] = PMG_MB_H;
__constant__ float C_ILO[NJ][6] = PMG_MB_ILO;
__constant__ float C_DSUM[NJ][3] = PMG_MB_DSUM;
__constant__ float C_SUBM[NJ][2] = PMG_MB_SUBM_MASS;
__constant__ float C_SUBC[NJ][2][3] = PMG_MB_SUBM_COM;
/* motor/limit row order of the non-contact constraint list (PMG_ROW_ORDER) */
__constant__ int C_ROWDOF[NJ] = {2, 3, 0, 1, 4, 7, 8, 5, 6};

/* ---------------------------------------------------------------- */
/* per-lane (= per movable link) model constants, staged ONCE per kernel in LDS
 * ([field][lane], conflict-free) and read on demand: keeps ~40 VGPRs free in
 * the 100-substep loop. */
constexpr int LC_RF = 0, LC_XYZ = 9, LC_AX = 12, LC_PRISM = 15, LC_MASS = 16, LC_H = 17, LC_ILO = 20, LC_DSUM = 26,
              LC_SM = 29, LC_SC = 31, LC_JLO = 37, LC_JHI = 38, LC_JDAMP = 39, LC_N = 40;
struct LaneTabStore { float t[LC_N][16]; };
struct LaneConst {
    const LaneTabStore* st;
    int col; /* min(lane, 8) */
    __device__ __forceinline__ float get(int f) const { return st->t[f][col]; }
    __device__ __forceinline__ float rf(int i) const { return get(LC_RF + i); }
    __device__ __forceinline__ float xyz(int i) const { return get(LC_XYZ + i); }
    __device__ __forceinline__ float ax(int i) const { return get(LC_AX + i); }
    __device__ __forceinline__ bool prismatic() const { return get(LC_PRISM) != 0.f; }
    __device__ __forceinline__ float mass() const { return get(LC_MASS); }
    __device__ __forceinline__ float h(int i) const { return get(LC_H + i); }
    __device__ __forceinline__ float ilo(int i) const { return get(LC_ILO + i); }
    __device__ __forceinline__ float dsum(int i) const { return get(LC_DSUM + i); }
    __device__ __forceinline__ float sm(int i) const { return get(LC_SM + i); }
    __device__ __forceinline__ float sc(int s, int i) const { return get(LC_SC + 3 * s + i); }
    __device__ __forceinline__ float jlo() const { return get(LC_JLO); }
    __device__ __forceinline__ float jhi() const { return get(LC_JHI); }
    __device__ __forceinline__ float jdamp() const { return get(LC_JDAMP); }
};

__device__ __forceinline__ void load_lane_const(LaneTabStore& st, LaneConst& c)
{
    int l = wv::lane();
    if (l < 16) {
        int m = l < NJ ? l : NJ - 1;
        for (int a = 0; a < 3; a++) {
            for (int b = 0; b < 3; b++) st.t[LC_RF + 3 * a + b][l] = C_JROT[m][a][b];
            st.t[LC_XYZ + a][l] = C_JXYZ[m][a];
            st.t[LC_AX + a][l] = C_JAXIS[m][a];
            st.t[LC_H + a][l] = C_H[m][a];
            st.t[LC_DSUM + a][l] = C_DSUM[m][a];
            st.t[LC_SC + a][l] = C_SUBC[m][0][a];
            st.t[LC_SC + 3 + a][l] = C_SUBC[m][1][a];
        }
        for (int a = 0; a < 6; a++) st.t[LC_ILO + a][l] = C_ILO[m][a];
        st.t[LC_PRISM][l] = (float)C_JTYPE[m];
        st.t[LC_MASS][l] = C_MASS[m];
        st.t[LC_SM][l] = C_SUBM[m][0];
        st.t[LC_SM + 1][l] = C_SUBM[m][1];
        st.t[LC_JLO][l] = C_JLO[m];
        st.t[LC_JHI][l] = C_JHI[m];
        st.t[LC_JDAMP][l] = C_JDAMP[m];
    }
    wv::lds_sync();
    c.st = &st;
    c.col = l < NJ ? l : NJ - 1;
}

/* ---------------------------------------------------------------- */
/* small vector helpers                                              */
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o)
{
    float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ float dot6(const float* a, const float* b) { return dot3(a, b) + dot3(a + 3, b + 3); }
/* o = R v, R row-major */
__device__ __forceinline__ void mat3v(const float* R, const float* v, float* o)
{
    float x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    float y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    float z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
__device__ __forceinline__ void mat3m(const float* A, const float* B, float* O)
{
    float t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) O[i] = t[i];
}
/* symmetric 3x3 (xx xy xz yy yz zz) times vector */
__device__ __forceinline__ void sym3v(const float* I, const float* v, float* o)
{
    float x = I[0] * v[0] + I[1] * v[1] + I[2] * v[2];
    float y = I[1] * v[0] + I[3] * v[1] + I[4] * v[2];
    float z = I[2] * v[0] + I[4] * v[1] + I[5] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
/* rigid inertia (m, H, Ibar) times motion vector -> force vector */
__device__ __forceinline__ void inertia_mul(const float* I10, const float* mv, float* f)
{
    float t[3], u[3];
    sym3v(I10 + 4, mv, t);
    cross3(I10 + 1, mv + 3, u); /* H x v */
    f[0] = t[0] + u[0]; f[1] = t[1] + u[1]; f[2] = t[2] + u[2];
    cross3(mv, I10 + 1, u);     /* w x H */
    f[3] = I10[0] * mv[3] + u[0]; f[4] = I10[0] * mv[4] + u[1]; f[5] = I10[0] * mv[5] + u[2];
}
/* motion x motion */
__device__ __forceinline__ void crm(const float* v, const float* m, float* o)
{
    float a[3], b[3], c[3];
    cross3(v, m, a);
    cross3(v, m + 3, b);
    cross3(v + 3, m, c);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
    o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
/* motion x* force */
__device__ __forceinline__ void crf(const float* v, const float* f, float* o)
{
    float a[3], b[3], c[3];
    cross3(v, f, a);
    cross3(v + 3, f + 3, b);
    cross3(v, f + 3, c);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
    o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}

/* inclusive sum over the joints on the path base -> this link (lanes 0..6 chain, 7/8 leaves of 6) */
__device__ __forceinline__ float chain_prefix(float x)
{
    int l = wv::lane();
    float y = l < 7 ? x : 0.f;
    y += wv::row_shr<1>(y, 0.f);
    y += wv::row_shr<2>(y, 0.f);
    y += wv::row_shr<4>(y, 0.f);
    float y6 = wv::bcast(y, 6);
    return l < 7 ? y : y6 + x;
}
/* inclusive sum over the subtree of this link */
__device__ __forceinline__ float chain_suffix(float x)
{
    int l = wv::lane();
    float x7 = wv::bcast(x, 7), x8 = wv::bcast(x, 8);
    float y = l < 6 ? x : (l == 6 ? (x + x7) + x8 : 0.f);
    y += wv::row_shl<1>(y, 0.f);
    y += wv::row_shl<2>(y, 0.f);
    y += wv::row_shl<4>(y, 0.f);
    return l < 7 ? y : x;
}

/* ---------------------------------------------------------------- */
struct Kin {
    float R[9], p[3]; /* world <- link frame */
    float S[6];       /* joint motion subspace */
};

/* one level of the affine-map scan: (R, p) <- (R', p') o (R, p) with (R', p') from lane i-S */
template <int S>
__device__ __forceinline__ void fk_scan_level(float* R, float* p)
{
    float A[9], a3[3];
#pragma unroll
    for (int a = 0; a < 9; a++) A[a] = wv::row_shr<S>(R[a], (a % 4 == 0) ? 1.f : 0.f);
#pragma unroll
    for (int a = 0; a < 3; a++) a3[a] = wv::row_shr<S>(p[a], 0.f);
    float t[3];
    mat3v(A, p, t);
    p[0] = a3[0] + t[0]; p[1] = a3[1] + t[1]; p[2] = a3[2] + t[2];
    mat3m(A, R, R);
}

/* forward kinematics of the chain; lanes >= 9 end up holding link 7's frame (lane 6's) */
__device__ __forceinline__ void fk(const LaneConst& c, float q, Kin& k)
{
    int l = wv::lane();
    float Lo[12]; /* local rotation (9) + offset (3) in the parent frame */
    float sq, cq;
    sincosf(q, &sq, &cq);
    float rf[9], ax[3];
#pragma unroll
    for (int a = 0; a < 9; a++) rf[a] = c.rf(a);
#pragma unroll
    for (int a = 0; a < 3; a++) ax[a] = c.ax(a);
    const bool prism = c.prismatic();
    if (prism) {
        float d[3];
        mat3v(rf, ax, d);
#pragma unroll
        for (int a = 0; a < 9; a++) Lo[a] = rf[a];
#pragma unroll
        for (int a = 0; a < 3; a++) Lo[9 + a] = c.xyz(a) + d[a] * q;
    } else { /* revolute about local z: L = Rfix * Rz(q) */
#pragma unroll
        for (int r = 0; r < 3; r++) {
            Lo[3 * r] = rf[3 * r] * cq + rf[3 * r + 1] * sq;
            Lo[3 * r + 1] = rf[3 * r + 1] * cq - rf[3 * r] * sq;
            Lo[3 * r + 2] = rf[3 * r + 2];
            Lo[9 + r] = c.xyz(r);
        }
    }
    /* inclusive scan of the affine maps (L, o) along lanes 0..7 with DPP row shifts:
     * T_i <- T_{i-s} o T_i for s = 1, 2, 4 (lanes without a source keep T_i: the
     * DPP fill is the identity map).  Lane 7 (finger 1) is a leaf of link 7 and
     * may take part; lane 8 (finger 2) hangs off lane 6 and is patched after.   */
    float R[9], p[3];
#pragma unroll
    for (int a = 0; a < 9; a++) R[a] = Lo[a];
#pragma unroll
    for (int a = 0; a < 3; a++) p[a] = Lo[9 + a];
    float own[12];
#pragma unroll
    for (int a = 0; a < 12; a++) own[a] = Lo[a];
    if (l == 8) { /* keep lane 8 out of the chain: identity */
#pragma unroll
        for (int a = 0; a < 9; a++) R[a] = (a % 4 == 0) ? 1.f : 0.f;
        p[0] = p[1] = p[2] = 0.f;
    }
    fk_scan_level<1>(R, p);
    fk_scan_level<2>(R, p);
    fk_scan_level<4>(R, p);
    {   /* lane 8 = T_6 o T_8local; lanes >= 9 = T_6 (tip-side consumers) */
        float A[9], a3[3];
#pragma unroll
        for (int a = 0; a < 9; a++) A[a] = wv::bcast(R[a], 6);
#pragma unroll
        for (int a = 0; a < 3; a++) a3[a] = wv::bcast(p[a], 6);
        if (l == 8) {
            float t[3];
            mat3v(A, own + 9, t);
            p[0] = a3[0] + t[0]; p[1] = a3[1] + t[1]; p[2] = a3[2] + t[2];
            mat3m(A, own, R);
        } else if (l >= NJ) {
#pragma unroll
            for (int a = 0; a < 9; a++) R[a] = A[a];
            p[0] = a3[0]; p[1] = a3[1]; p[2] = a3[2];
        }
    }
#pragma unroll
    for (int a = 0; a < 9; a++) k.R[a] = R[a];
#pragma unroll
    for (int a = 0; a < 3; a++) k.p[a] = p[a];
    float aw[3];
    mat3v(R, ax, aw);
    if (prism) {
        k.S[0] = k.S[1] = k.S[2] = 0.f;
        k.S[3] = aw[0]; k.S[4] = aw[1]; k.S[5] = aw[2];
    } else {
        k.S[0] = aw[0]; k.S[1] = aw[1]; k.S[2] = aw[2];
        cross3(p, aw, k.S + 3);
    }
    if (l >= NJ) {
#pragma unroll
        for (int a = 0; a < 6; a++) k.S[a] = 0.f;
    }
}

/* tip frame (uniform): tip position and the rotation of link 7 */
__device__ __forceinline__ void tip_frame(const Kin& k, float* tip, float* Rt)
{
    float t[12];
#pragma unroll
    for (int a = 0; a < 3; a++) t[a] = k.p[a] + k.R[3 * a + 2] * TIP_Z;
#pragma unroll
    for (int a = 0; a < 9; a++) t[3 + a] = k.R[a];
    float o[12];
    wv::bcastn<12>(t, 6, o);
#pragma unroll
    for (int a = 0; a < 3; a++) tip[a] = o[a];
#pragma unroll
    for (int a = 0; a < 9; a++) Rt[a] = o[3 + a];
}

/* ---------------------------------------------------------------- */
/* inverse kinematics: damped least squares on the 6 x 7 tip Jacobian,
 * dq = J^T (J J^T + 0.5 I)^-1 e  ( == (J^T J + 0.5 I)^-1 J^T e of BussIK's
 * CalcDeltaThetasDLS2), <= 40 iterations, stop on position residual.      */
/* [BULLET-PRIOR] btMatrix3x3::getRotation; the three "largest diagonal" cases are written out
 * with constant indices so that nothing is dynamically indexed (no scratch in the IK loop) */
__device__ __forceinline__ void quat_case(const float* m, int i, int j, int k, float* q)
{
    float s = sqrtf(m[4 * i] - m[4 * j] - m[4 * k] + 1.f);
    q[i] = 0.5f * s;
    s = 0.5f / s;
    q[3] = (m[3 * k + j] - m[3 * j + k]) * s;
    q[j] = (m[3 * j + i] + m[3 * i + j]) * s;
    q[k] = (m[3 * k + i] + m[3 * i + k]) * s;
}
__device__ __forceinline__ void quat_from_R(const float* m, float* q)
{
    float tr = m[0] + m[4] + m[8];
    if (tr > 0.f) {
        float s = sqrtf(tr + 1.f);
        q[3] = 0.5f * s;
        s = 0.5f / s;
        q[0] = (m[7] - m[5]) * s; q[1] = (m[2] - m[6]) * s; q[2] = (m[3] - m[1]) * s;
    } else if (m[0] < m[4]) {
        if (m[4] < m[8]) quat_case(m, 2, 0, 1, q); else quat_case(m, 1, 2, 0, q);
    } else {
        if (m[0] < m[8]) quat_case(m, 2, 0, 1, q); else quat_case(m, 0, 1, 2, q);
    }
}

/* solve the SPD system A y = b in place, A given by its 21 upper entries (row-major upper) */
__device__ __forceinline__ void spd6_solve(float (*A)[6], float* b)
{
    /* Cholesky A = L L^T, L stored in the lower triangle */
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j][k] * A[j][k];
        float inv = rsqrtf(d);
        A[j][j] = inv; /* store 1/L_jj */
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            float s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i][k] * A[j][k];
            A[i][j] = s * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= A[i][k] * b[k];
        b[i] = s * A[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        float s = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= A[k][i] * b[k];
        b[i] = s * A[i][i];
    }
}

__device__ __forceinline__ float ik_solve(const LaneConst& c, float q, const float* target)
{
    int l = wv::lane();
    float diff = 1e30f;
    for (int it = 0; it < IK_MAX_ITER && diff > IK_THRESHOLD; it++) {
        Kin k;
        fk(c, q, k);
        float tip[3], Rt[9];
        tip_frame(k, tip, Rt);
        float e[6];
#pragma unroll
        for (int a = 0; a < 3; a++) e[a] = target[a] - tip[a];
        diff = sqrtf(dot3(e, e));
        /* orientation error towards the fixed tool quaternion [0,-1,0,0] (kuka.py:42):
         * dq = tq * conj(sq) with tq = (0,-1,0,0)  ->  angle-axis */
        float sq[4];
        quat_from_R(Rt, sq);
        /* tq * (-sx,-sy,-sz,sw), tq = (x=0,y=-1,z=0,w=0) */
        float dx = sq[2], dy = -sq[3], dz = -sq[0], dw = -sq[1];
        float vn = sqrtf(dx * dx + dy * dy + dz * dz);
        float ang = 2.f * atan2f(vn, dw);
        if (ang > 3.14159265358979f) ang -= 6.28318530717959f;
        float sc = vn > 1e-12f ? ang / vn : 0.f;
        e[3] = dx * sc; e[4] = dy * sc; e[5] = dz * sc;
        /* Jacobian column of this lane's joint */
        float col[6] = {0, 0, 0, 0, 0, 0};
        if (l < 7) {
            float r[3] = {tip[0] - k.p[0], tip[1] - k.p[1], tip[2] - k.p[2]};
            cross3(k.S, r, col);
            col[3] = k.S[0]; col[4] = k.S[1]; col[5] = k.S[2];
        }
        float A[6][6];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) {
                float s = wv::row_sum(col[a] * col[b]);
                if (a == b) s += IK_DAMPING;
                A[a][b] = s;
                A[b][a] = s;
            }
        spd6_solve(A, e);
        float dq = l < 7 ? dot6(col, e) : 0.f;
        float mx = wv::row_max(fabsf(dq));
        if (mx > IK_MAX_STEP) dq *= IK_MAX_STEP / mx;
        q += dq;
    }
    return q;
}

/* ---------------------------------------------------------------- */
/* robot forward dynamics pieces                                      */
struct Dyn {
    float minv[NJ]; /* this lane's row of M^-1 */
    float v[6];     /* spatial velocity of this link */
};

/* world-frame 10-parameter inertia of this lane's merged body */
__device__ __forceinline__ void body_inertia(const LaneConst& c, const Kin& k, float* I10)
{
    float hw[3], hl[3] = {c.h(0), c.h(1), c.h(2)}, il[6];
#pragma unroll
    for (int a = 0; a < 6; a++) il[a] = c.ilo(a);
    mat3v(k.R, hl, hw);
    const float* p = k.p;
    float m = c.mass();
    I10[0] = m;
    I10[1] = m * p[0] + hw[0]; I10[2] = m * p[1] + hw[1]; I10[3] = m * p[2] + hw[2];
    /* R Ilo R^T */
    const float* R = k.R;
    float T[9];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        T[3 * r] = R[3 * r] * il[0] + R[3 * r + 1] * il[1] + R[3 * r + 2] * il[2];
        T[3 * r + 1] = R[3 * r] * il[1] + R[3 * r + 1] * il[3] + R[3 * r + 2] * il[4];
        T[3 * r + 2] = R[3 * r] * il[2] + R[3 * r + 1] * il[4] + R[3 * r + 2] * il[5];
    }
    float I[6];
    I[0] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2];
    I[1] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5];
    I[2] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
    I[3] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5];
    I[4] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8];
    I[5] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
    /* shift link origin -> world origin: m[(p.p)1 - p p^T] + [2(p.hw)1 - p hw^T - hw p^T] */
    float pp = dot3(p, p), ph = dot3(p, hw);
    float dg = m * pp + 2.f * ph;
    I10[4] = I[0] + dg - m * p[0] * p[0] - 2.f * p[0] * hw[0];
    I10[5] = I[1] - m * p[0] * p[1] - p[0] * hw[1] - hw[0] * p[1];
    I10[6] = I[2] - m * p[0] * p[2] - p[0] * hw[2] - hw[0] * p[2];
    I10[7] = I[3] + dg - m * p[1] * p[1] - 2.f * p[1] * hw[1];
    I10[8] = I[4] - m * p[1] * p[2] - p[1] * hw[2] - hw[1] * p[2];
    I10[9] = I[5] + dg - m * p[2] * p[2] - 2.f * p[2] * hw[2];
}

/* M^-1 (this lane's row) by CRBA + in-place Gauss-Jordan; also returns own inertia */
__device__ __forceinline__ void mass_inverse(const Kin& k, const float* I10, float* minv)
{
    int l = wv::lane();
    float Ic[10];
#pragma unroll
    for (int a = 0; a < 10; a++) Ic[a] = chain_suffix(I10[a]);
    float SF[12];
#pragma unroll
    for (int a = 0; a < 6; a++) SF[a] = k.S[a];
    inertia_mul(Ic, k.S, SF + 6); /* F = Ic S */
    float a[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        float o[12];
        wv::bcastn<12>(SF, j, o);
        bool related = !((l == 7 && j == 8) || (l == 8 && j == 7));
        float lo = dot6(o, SF + 6); /* S_j . F_l   (j ancestor-or-self of l) */
        float hi = dot6(SF, o + 6); /* S_l . F_j   (j descendant of l)        */
        a[j] = related ? (j <= l ? lo : hi) : 0.f;
    }
    if (l >= NJ) { /* keep idle lanes finite: identity rows */
#pragma unroll
        for (int j = 0; j < NJ; j++) a[j] = 0.f;
    }
    /* in-place Gauss-Jordan inversion, lane = row, no pivoting (SPD) */
#pragma unroll
    for (int p = 0; p < NJ; p++) {
        float piv[NJ];
        wv::bcastn<NJ>(a, p, piv);
        float d = 1.f / piv[p];
        float f = a[p];
        if (l == p) {
#pragma unroll
            for (int j = 0; j < NJ; j++) a[j] = piv[j] * d;
            a[p] = d;
        } else {
            float fd = f * d;
#pragma unroll
            for (int j = 0; j < NJ; j++) a[j] -= fd * piv[j];
            a[p] = -fd;
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) minv[j] = a[j];
}

/* bias torque h_l = S_l . sum_{subtree} (I a_b + v x* I v - f_ext) with gravity and Bullet's link damping */
__device__ __forceinline__ float bias_torque(const LaneConst& c, const Kin& k, const float* I10, float qd)
{
    float s[6], v[6], cb[6], ab[6];
#pragma unroll
    for (int a = 0; a < 6; a++) s[a] = k.S[a] * qd;
#pragma unroll
    for (int a = 0; a < 6; a++) v[a] = chain_prefix(s[a]);
    crm(v, s, cb);
#pragma unroll
    for (int a = 0; a < 6; a++) ab[a] = chain_prefix(cb[a]);
    float f[6], hm[6], t[6];
    inertia_mul(I10, ab, f);
    inertia_mul(I10, v, hm);
    crf(v, hm, t);
#pragma unroll
    for (int a = 0; a < 6; a++) f[a] += t[a];
    /* gravity: force (0,0,-m g) at the COM -> [H x g ; m g] */
    f[0] -= -GRAVITY * I10[2];
    f[1] -= GRAVITY * I10[1];
    f[5] -= -GRAVITY * I10[0];
    /* link damping, angular: -(R Dsum R^T w) k (1+|w|) */
    {
        float wl[3], dw[3];
        const float* R = k.R;
        wl[0] = (R[0] * v[0] + R[3] * v[1] + R[6] * v[2]) * c.dsum(0);
        wl[1] = (R[1] * v[0] + R[4] * v[1] + R[7] * v[2]) * c.dsum(1);
        wl[2] = (R[2] * v[0] + R[5] * v[1] + R[8] * v[2]) * c.dsum(2);
        mat3v(R, wl, dw);
        float ka = LINK_DAMPING * (1.f + sqrtf(dot3(v, v)));
        f[0] += dw[0] * ka; f[1] += dw[1] * ka; f[2] += dw[2] * ka;
    }
    /* link damping, linear, per massive sub-body at its own COM */
#pragma unroll
    for (int sb = 0; sb < 2; sb++) {
        float cw[3], vc[3], fd[3], nd[3], scl[3] = {c.sc(sb, 0), c.sc(sb, 1), c.sc(sb, 2)};
        mat3v(k.R, scl, cw);
        cw[0] += k.p[0]; cw[1] += k.p[1]; cw[2] += k.p[2];
        cross3(v, cw, vc);
        vc[0] += v[3]; vc[1] += v[4]; vc[2] += v[5];
        float kl = c.sm(sb) * LINK_DAMPING * (1.f + sqrtf(dot3(vc, vc)));
        fd[0] = vc[0] * kl; fd[1] = vc[1] * kl; fd[2] = vc[2] * kl;
        cross3(cw, fd, nd);
        f[0] += nd[0]; f[1] += nd[1]; f[2] += nd[2];
        f[3] += fd[0]; f[4] += fd[1]; f[5] += fd[2];
    }
    if (wv::lane() >= NJ) {
#pragma unroll
        for (int a = 0; a < 6; a++) f[a] = 0.f;
    }
    float fc[6];
#pragma unroll
    for (int a = 0; a < 6; a++) fc[a] = chain_suffix(f[a]);
    return dot6(k.S, fc);
}

/* ---------------------------------------------------------------- */
/* non-contact constraint rows (joint motors + joint limits).  Lane d owns the
 * motor row and the (optional) limit row of DoF d: their right-hand sides,
 * 1/diag, impulse bounds and accumulated impulses live in that lane's
 * registers; a visit broadcasts only the resulting impulse change. */
struct NcRows {
    float dinv, den;     /* 1 / (M^-1)_dd and (M^-1)_dd */
    float rhs_m, rhs_l;  /* motor / limit right-hand sides */
    float imp_m;         /* motor max impulse (0 = disabled) */
    float app_m, app_l;  /* accumulated impulses */
    float prev_m, prev_l;/* their values at the start of the current iteration (residual test) */
    float sg_l;          /* limit Jacobian sign: +1 lower, -1 upper */
    unsigned mot_active; /* wave-uniform bit masks over DoFs */
    unsigned lim_active;
};

__device__ __forceinline__ void build_nc_rows(const LaneConst& c, const float* minv, float q, float qd, float mtarget,
                                              float mimp, NcRows& r)
{
    int l = wv::lane() < NJ ? wv::lane() : 0;
    float den = minv[0];
#pragma unroll
    for (int j = 1; j < NJ; j++) den = (l == j) ? minv[j] : den;
    r.den = den > SIMD_EPS ? den : 0.f;
    r.dinv = den > SIMD_EPS ? 1.f / den : 0.f;
    /* btMultiBodyJointMotor: target velocity kp*(q*-q)/dt + qd + kd*(0-qd) */
    float tv = ARM_KP * (mtarget - q) / DT + qd + ARM_KD * (0.f - qd);
    r.rhs_m = (tv - qd) * r.dinv;
    r.imp_m = mimp;
    /* btMultiBodyJointLimitConstraint: active when (q-lo) <= 0 or (hi-q) <= 0 */
    float plo = q - c.jlo(), phi = c.jhi() - q;
    bool alo = plo <= 0.f, ahi = !alo && phi <= 0.f;
    float pen = alo ? plo : phi;
    r.sg_l = alo ? 1.f : -1.f;
    r.rhs_l = (-pen * JOINT_ERP / DT - r.sg_l * qd) * r.dinv;
    r.app_m = r.app_l = r.prev_m = r.prev_l = 0.f;
    bool valid = wv::lane() < NJ;
    r.mot_active = (unsigned)wv::ballot(valid && mimp > 0.f) & 0x1FFu;
    r.lim_active = (unsigned)wv::ballot(valid && (alo || ahi)) & 0x1FFu;
}

/* squared velocity change of the rows this lane owns during the iteration just finished
 * (each row is visited once per iteration, so it is its impulse change times (M^-1)_dd) */
__device__ __forceinline__ float nc_residual(NcRows& r)
{
    float a = (r.app_m - r.prev_m) * r.den, b = (r.app_l - r.prev_l) * r.den;
    r.prev_m = r.app_m;
    r.prev_l = r.app_l;
    return fmaxf(a * a, b * b);
}

/* one Gauss-Seidel visit of the motor (KIND 0) or limit (KIND 1) row of DoF D:
 * every lane evaluates "its own" row in SIMD, lane D's impulse change is broadcast */
template <int D, int KIND>
__device__ __forceinline__ void nc_row_solve(NcRows& r, const float* minv, float& dqd)
{
    if (!(((KIND == 0 ? r.mot_active : r.lim_active) >> D) & 1u)) return;
    float sg = KIND == 0 ? 1.f : r.sg_l;
    float rhs = KIND == 0 ? r.rhs_m : r.rhs_l;
    float lo = KIND == 0 ? -r.imp_m : 0.f, hi = KIND == 0 ? r.imp_m : LIMIT_MAX_IMPULSE;
    float app = KIND == 0 ? r.app_m : r.app_l;
    float sum = app + (rhs - sg * dqd * r.dinv);
    float napp = __builtin_amdgcn_fmed3f(sum, lo, hi);
    float sdelta = sg * (napp - app);          /* signed impulse change mapped to joint space */
    float d = wv::bcast(sdelta, D);
    bool mine = wv::lane() == D;
    if (KIND == 0) r.app_m = mine ? napp : r.app_m; else r.app_l = mine ? napp : r.app_l;
    dqd += minv[D] * d;
}

/* rows in list order: motors (dof 2,3,0,1,4,7,8,5,6) then limits (same dof order);
 * Bullet walks the list backwards on even iterations */
__device__ __forceinline__ void nc_sweep(NcRows& r, bool forward, const float* minv, float& dqd)
{
    if (forward) {
        nc_row_solve<2, 0>(r, minv, dqd); nc_row_solve<3, 0>(r, minv, dqd); nc_row_solve<0, 0>(r, minv, dqd);
        nc_row_solve<1, 0>(r, minv, dqd); nc_row_solve<4, 0>(r, minv, dqd); nc_row_solve<7, 0>(r, minv, dqd);
        nc_row_solve<8, 0>(r, minv, dqd); nc_row_solve<5, 0>(r, minv, dqd); nc_row_solve<6, 0>(r, minv, dqd);
        if (r.lim_active) {
            nc_row_solve<2, 1>(r, minv, dqd); nc_row_solve<3, 1>(r, minv, dqd); nc_row_solve<0, 1>(r, minv, dqd);
            nc_row_solve<1, 1>(r, minv, dqd); nc_row_solve<4, 1>(r, minv, dqd); nc_row_solve<7, 1>(r, minv, dqd);
            nc_row_solve<8, 1>(r, minv, dqd); nc_row_solve<5, 1>(r, minv, dqd); nc_row_solve<6, 1>(r, minv, dqd);
        }
    } else {
        if (r.lim_active) {
            nc_row_solve<6, 1>(r, minv, dqd); nc_row_solve<5, 1>(r, minv, dqd); nc_row_solve<8, 1>(r, minv, dqd);
            nc_row_solve<7, 1>(r, minv, dqd); nc_row_solve<4, 1>(r, minv, dqd); nc_row_solve<1, 1>(r, minv, dqd);
            nc_row_solve<0, 1>(r, minv, dqd); nc_row_solve<3, 1>(r, minv, dqd); nc_row_solve<2, 1>(r, minv, dqd);
        }
        nc_row_solve<6, 0>(r, minv, dqd); nc_row_solve<5, 0>(r, minv, dqd); nc_row_solve<8, 0>(r, minv, dqd);
        nc_row_solve<7, 0>(r, minv, dqd); nc_row_solve<4, 0>(r, minv, dqd); nc_row_solve<1, 0>(r, minv, dqd);
        nc_row_solve<0, 0>(r, minv, dqd); nc_row_solve<3, 0>(r, minv, dqd); nc_row_solve<2, 0>(r, minv, dqd);
    }
}

}  // namespace pmg
#endif
