/*
 * pmg_oracle.c -- CPU restatement of the reference's per-step hot path.
 * TEST INFRASTRUCTURE ONLY (see pmg_oracle.h: "PARITY UNPINNED").
 *
 * What is restated, and from where (P/ = pybullet_multigoal_gym/ in the reference):
 *   env orchestration ... P/envs/base_envs/base_env.py:124-138,203-220
 *   robot ................ P/robots/kuka.py:27,35-51,120-165,167-225,227-301
 *   single-step tasks .... P/envs/base_envs/kuka_single_step_base_env.py:48-56,76-148,193-244
 *                          P/envs/task_envs/kuka_single_step_envs.py:4-59
 *   block stack .......... P/envs/base_envs/kuka_multi_step_base_env.py:62-81,183-250,255-345
 *                          P/envs/task_envs/kuka_multi_step_envs.py:34-87
 *   RNG .................. gym 0.17.3 seeding.np_random + numpy RandomState (MT19937)
 *   physics .............. pybullet~=3.0.6 (NOT in the container): Bullet's
 *       btMultiBody articulated-body algorithm, btMultiBodyConstraintSolver
 *       (projected Gauss-Seidel over joint motors, joint limits, contact and
 *       friction rows), box-box SAT/clipping contacts and BussIK damped
 *       least squares IK, restated from the published algorithm.  Every such
 *       statement is tagged [BULLET-PRIOR]; DESIGN.md lists them.
 */
#include "pmg_oracle.h"
#include "../include/pmg_model.h"
#include "../include/pmg_sha512_const.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef PMGO_FLOAT
typedef float real;
#define RSQRT sqrtf
#define RFABS fabsf
#define RSIN sinf
#define RCOS cosf
#define RACOS acosf
#else
typedef double real;
#define RSQRT sqrt
#define RFABS fabs
#define RSIN sin
#define RCOS cos
#define RACOS acos
#endif

/* ------------------------------------------------------------------ */
/* [BULLET-PRIOR] choices, switchable (oracle only; the product compiles the defaults in).  With no PyBullet to ask,
 * each of these was settled from Bullet 3.0.x's published behaviour; pmgo_set_prior() lets the first real capture
 * (tools/gen_reference_fixtures.py --real) rank the alternatives instead of starting a debugging session.
 * Structural choices: the contact POINTS are regenerated every substep (Bullet refreshes a persistent 4-point manifold:
 * for the box pairs of this scene btBoxBoxDetector re-adds the same clipped face points each substep, so the point sets
 * agree while a contact persists); what a persistent manifold additionally carries -- the cached impulses -- is the
 * `contact_warm_start` switch below.  The SAT cylinder pairs (Bullet: GJK/EPA) are cross-checked against an independent
 * closest-point computation in tests/test_oracle_physics.py.                                                 */
/* ------------------------------------------------------------------ */
enum { PRIOR_MOTOR_IMPULSE_DT, PRIOR_LINK_DAMPING, PRIOR_JOINT_ERP, PRIOR_CONTACT_MARGIN, PRIOR_RESIDUAL_THRESHOLD,
       PRIOR_IK_DAMPING, PRIOR_LINEAR_SLOP, PRIOR_WARM_START, PRIOR_DAMPING_PER_SUBSTEP, PRIOR_FRICTION_DIRS,
       PRIOR_SOLVER_ITERATIONS, PRIOR_CONTACT_WARM_START, PRIOR_STATE_F32, PRIOR_N };
static const char* const PRIOR_NAME[PRIOR_N] = {
    "motor_impulse_dt",     /* 0.04: max motor impulse = force x fixedTimeStep; 0.002 = force x substep */
    "link_damping",         /* 0.04: btMultiBody linear / angular damping; 0 = none */
    "joint_erp",            /* 0.2: joint-limit error reduction */
    "contact_margin",       /* 0.002: speculative contact distance of the regenerated manifold */
    "residual_threshold",   /* 1e-7: solver early exit; 0 = always all iterations */
    "ik_damping",           /* 0.5: DLS damping of calculateInverseKinematics */
    "linear_slop",          /* 1e-5 */
    "warm_start",           /* 0: non-contact rows start each substep from zero; f in (0, 1] = from f x last substep's impulses */
    "damping_per_substep",  /* 0: URDF joint damping latched once per stepSimulation; 1 = re-evaluated every substep */
    "friction_dirs",        /* 2: two btPlaneSpace1 friction rows per contact; 1 = the first of them only */
    "solver_iterations",    /* 5: base_env.py:37,218 numSolverIterations -- NOT a prior (the reference sets it); switchable
                             * only to measure how much of a behaviour (creep of resting stacks / held blocks) is the
                             * 5-iteration truncation of Gauss-Seidel */
    "contact_warm_start",   /* 0: contact impulses start each substep from zero (what btMultiBodyConstraintSolver::
                             * setupMultiBodyContactConstraint does: its warm-starting branch is compiled out for
                             * multibody contacts, and every body PyBullet loads from a URDF is a btMultiBody);
                             * f in (0, 1]: Bullet's persistent-manifold behaviour for rigid bodies -- a contact point that
                             * persists (same pair, within the 0.02 breaking threshold of a cached point) starts from
                             * f x its last normal impulse (btSequentialImpulseConstraintSolver: f = 0.85) */
    "state_f32_per_substep" /* 0.  NOT a prior, a yardstick: 1 = joint / block / door positions and velocities are rounded to
                             * float32 at the end of every substep while all arithmetic stays float64 -- the best any
                             * implementation that KEEPS ITS STATE in float32 can do.  Its deviation from the plain float64
                             * oracle is the chaos floor of tools/teacher_forced.py (how often a single env step bifurcates
                             * under float32-sized state noise, whatever the arithmetic) */
};
static double G_PRIOR[PRIOR_N] = {0.04, 0.04, 0.2, 0.002, 1e-7, 0.5, 1e-5, 0.0, 0.0, 2.0, 5.0, 0.0, 0.0};
static const double PRIOR_DEFAULT[PRIOR_N] = {0.04, 0.04, 0.2, 0.002, 1e-7, 0.5, 1e-5, 0.0, 0.0, 2.0, 5.0, 0.0, 0.0};

/* ------------------------------------------------------------------ */
/* constants                                    */
/* ------------------------------------------------------------------ */
#define NJ PMG_NJ
#define NL PMG_BL_N
#define NBMAX 5
#define GRAVITY ((real)9.81)           /* base_env.py:17,215 */
#define SUBSTEP_DT ((real)0.002)       /* base_env.py:17,217: fixedTimeStep 0.04 / numSubSteps 20 */
#define SUBSTEPS 20                    /* base_env.py:17,219 */
#define SIM_STEPS 5                    /* kuka.py:223-225 */
#define SOLVER_ITERS 5                 /* base_env.py:37,218 */
#define PHYSICS_DT ((real)G_PRIOR[PRIOR_MOTOR_IMPULSE_DT]) /* base_env.py:217 (m_physicsDeltaTime): clamps motor impulses */
#define CONTACT_ERP ((real)0.9)        /* base_env.py:216 setDefaultContactERP */
#define JOINT_ERP ((real)G_PRIOR[PRIOR_JOINT_ERP]) /* [BULLET-PRIOR] btContactSolverInfo::m_erp default 0.2 */
#define LINEAR_SLOP ((real)G_PRIOR[PRIOR_LINEAR_SLOP]) /* [BULLET-PRIOR] PyBullet createEmptyDynamicsWorld: 1e-5 */
#define RESIDUAL_THRESHOLD ((real)G_PRIOR[PRIOR_RESIDUAL_THRESHOLD]) /* [BULLET-PRIOR] m_leastSquaresResidualThreshold 1e-7 */
#define LINK_DAMPING ((real)G_PRIOR[PRIOR_LINK_DAMPING]) /* [BULLET-PRIOR] btMultiBody m_linearDamping / m_angularDamping 0.04 */
#define LIMIT_MAX_IMPULSE ((real)100.0)/* [BULLET-PRIOR] btMultiBodyConstraint m_maxAppliedImpulse */
#define ARM_KP ((real)0.03)            /* kuka.py:289 */
#define ARM_KD ((real)1.0)             /* kuka.py:290 */
#define ARM_FORCE ((real)200.0)        /* kuka.py:288 */
#define FINGER_FORCE ((real)50.0)      /* kuka.py:299 */
#define FINGER_LIMIT ((real)0.035)     /* kuka.py:71 */
#define IK_MAX_ITER 40                 /* kuka.py:278 */
#define IK_THRESHOLD ((real)1e-5)      /* kuka.py:279 */
#define IK_DAMPING ((real)G_PRIOR[PRIOR_IK_DAMPING]) /* [BULLET-PRIOR] default joint_damping in calculateInverseKinematics: 0.5 */
#define IK_MAX_STEP ((real)(45.0 * 3.14159265358979323846 / 180.0)) /* [BULLET-PRIOR] MaxAngleDLS */
#define CONTACT_MARGIN ((real)G_PRIOR[PRIOR_CONTACT_MARGIN]) /* build choice: speculative-contact distance 0.002 */
#define GBASE_FRICTION ((real)0.5)     /* [BULLET-PRIOR] btCollisionObject default friction (no <contact> tag) */
#define EDGE_FUDGE ((real)1.05)        /* [BULLET-PRIOR] btBoxBoxDetector fudge_factor */
#define MAX_CONTACTS 64
#define PI_R ((real)3.14159265358979323846)

static const int JPARENT_BL[NL] = PMG_BL_PARENT;
static const int BL_TYPE[NL] = PMG_BL_TYPE;
static const int BL_DOF[NL] = PMG_BL_DOF;
static const double BL_XYZ[NL][3] = PMG_BL_XYZ;
static const double BL_ROT[NL][3][3] = PMG_BL_ROT;
static const double BL_AXIS[NL][3] = PMG_BL_AXIS;
static const double BL_MASS[NL] = PMG_BL_MASS;
static const double BL_COM[NL][3] = PMG_BL_COM;
static const double BL_INERTIA[NL][3] = PMG_BL_INERTIA;
static const double JLO[NJ] = PMG_JLO;
static const double JHI[NJ] = PMG_JHI;
static const double JDAMP[NJ] = PMG_JDAMP;
static const int ROW_ORDER[18] = PMG_ROW_ORDER;
static const double FINGER_HALF[3] = PMG_FINGER_HALF;
static const double TABLE_HALF[3] = PMG_TABLE_HALF;
/* The robot URDF's base link (iiwa14_parallel_jaw.urdf:37-58): a 5 x 5 x 0.002 m box under link_0 at the world origin, lateral
 * friction 1 -- the floor.  An object knocked off the table lands on it (z = 0.001 + its half height) instead of falling for
 * the rest of the episode.  One static pair per object serves both: below PLANE_SWITCH_Z (the object's bounding sphere -- cube
 * 0.026, puck 0.0317 -- can reach the floor; it is then 8.6 cm under the table top and can touch the table at its side walls only)
 * the static partner of the object is the floor, above it the table.  Build choice, documented in DESIGN.md: an object that
 * already lies on the floor does not collide with the table's side walls. */
static const double PLANE_HALF[3] = PMG_PLANE_HALF;
#define PLANE_SWITCH_Z 0.04
static const double BLOCK_HALF[3] = PMG_BLOCK_HALF;
static const double BLOCK_INERTIA[3] = PMG_BLOCK_INERTIA;

/* kuka.py:27 */
static const double REST_POSE0[7] = {0, -0.5592432, 0, 1.733180, 0, -0.8501557, 0};
/* kuka.py:42 (xyzw) */
static const double TOOL_QUAT[4] = {0, -1, 0, 0};

/* ------------------------------------------------------------------ */
/* small linear algebra                                                */
/* ------------------------------------------------------------------ */
static inline void v3set(real* a, real x, real y, real z) { a[0] = x; a[1] = y; a[2] = z; }
static inline void v3cpy(real* a, const real* b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static inline void v3add(real* o, const real* a, const real* b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3sub(real* o, const real* a, const real* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3axpy(real* o, real s, const real* a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static inline real v3dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(real* o, const real* a, const real* b)
{
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline real v3norm(const real* a) { return RSQRT(v3dot(a, a)); }
/* o = R v (R row-major 3x3) */
static inline void m3v(real* o, const real* R, const real* v)
{
    real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    real y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    real z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tv(real* o, const real* R, const real* v)
{
    real x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    real y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    real z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3m(real* o, const real* A, const real* B)
{
    real t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(o, t, sizeof(t));
}
static void quat_to_R(const real* q, real* R) /* q = xyzw */
{
    real x = q[0], y = q[1], z = q[2], w = q[3];
    real d = x * x + y * y + z * z + w * w;
    real s = (real)2.0 / d;
    real xs = x * s, ys = y * s, zs = z * s;
    real wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    R[0] = 1 - (yy + zz); R[1] = xy - wz; R[2] = xz + wy;
    R[3] = xy + wz; R[4] = 1 - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy; R[7] = yz + wx; R[8] = 1 - (xx + yy);
}
/* [BULLET-PRIOR] btMatrix3x3::getRotation (Shepperd) */
static void R_to_quat(const real* m, real* q)
{
    real tr = m[0] + m[4] + m[8];
    if (tr > 0) {
        real s = RSQRT(tr + 1);
        q[3] = s * (real)0.5; s = (real)0.5 / s;
        q[0] = (m[7] - m[5]) * s; q[1] = (m[2] - m[6]) * s; q[2] = (m[3] - m[1]) * s;
    } else {
        int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
        int j = (i + 1) % 3, k = (i + 2) % 3;
        real s = RSQRT(m[4 * i] - m[4 * j] - m[4 * k] + 1);
        real t[4];
        t[i] = s * (real)0.5; s = (real)0.5 / s;
        t[3] = (m[3 * k + j] - m[3 * j + k]) * s;
        t[j] = (m[3 * j + i] + m[3 * i + j]) * s;
        t[k] = (m[3 * k + i] + m[3 * i + k]) * s;
        q[0] = t[0]; q[1] = t[1]; q[2] = t[2]; q[3] = t[3];
    }
}
static void quat_mul(real* o, const real* a, const real* b) /* xyzw, o = a*b */
{
    real x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    real y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    real z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    real w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* ------------------------------------------------------------------ */
/* SHA-512, MT19937, gym seeding, RandomState draws                     */
/* ------------------------------------------------------------------ */
static const uint64_t SHA_K[80] = PMG_SHA512_K;
static const uint64_t SHA_H0[8] = PMG_SHA512_H0;
#define ROTR64(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
static void sha512(const uint8_t* msg, size_t len, uint8_t out[64])
{
    uint64_t h[8];
    memcpy(h, SHA_H0, sizeof(h));
    size_t total = ((len + 17 + 127) / 128) * 128;
    uint8_t* buf = (uint8_t*)calloc(total, 1);
    memcpy(buf, msg, len);
    buf[len] = 0x80;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; i++) buf[total - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (size_t off = 0; off < total; off += 128) {
        uint64_t w[80];
        for (int i = 0; i < 16; i++) {
            uint64_t v = 0;
            for (int b = 0; b < 8; b++) v = (v << 8) | buf[off + 8 * i + b];
            w[i] = v;
        }
        for (int i = 16; i < 80; i++) {
            uint64_t s0 = ROTR64(w[i - 15], 1) ^ ROTR64(w[i - 15], 8) ^ (w[i - 15] >> 7);
            uint64_t s1 = ROTR64(w[i - 2], 19) ^ ROTR64(w[i - 2], 61) ^ (w[i - 2] >> 6);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint64_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 80; i++) {
            uint64_t S1 = ROTR64(e, 14) ^ ROTR64(e, 18) ^ ROTR64(e, 41);
            uint64_t ch = (e & f) ^ (~e & g);
            uint64_t t1 = hh + S1 + ch + SHA_K[i] + w[i];
            uint64_t S0 = ROTR64(a, 28) ^ ROTR64(a, 34) ^ ROTR64(a, 39);
            uint64_t mj = (a & b) ^ (a & c) ^ (b & c);
            uint64_t t2 = S0 + mj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    free(buf);
    for (int i = 0; i < 8; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(h[i] >> (56 - 8 * b));
}

typedef struct {
    uint32_t mt[624];
    int idx;
} mt19937;

static void mt_init_genrand(mt19937* s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253U * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}
static void mt_init_by_array(mt19937* s, const uint32_t* key, int klen)
{
    mt_init_genrand(s, 19650218U);
    int i = 1, j = 0;
    int k = 624 > klen ? 624 : klen;
    for (; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
        if (j >= klen) j = 0;
    }
    for (k = 623; k; k--) {
        s->mt[i] = (s->mt[i] ^ ((s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
        i++;
        if (i >= 624) { s->mt[0] = s->mt[623]; i = 1; }
    }
    s->mt[0] = 0x80000000U;
    s->idx = 624;
}
static uint32_t mt_next(mt19937* s)
{
    if (s->idx >= 624) {
        uint32_t* mt = s->mt;
        int kk;
        for (kk = 0; kk < 624 - 397; kk++) {
            uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        for (; kk < 623; kk++) {
            uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}
/* numpy RandomState.random_sample */
static double mt_double(mt19937* s)
{
    uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}
/* numpy RandomState.uniform(low, high): low + (high-low)*random_sample() */
static double mt_uniform(mt19937* s, double lo, double hi) { return lo + (hi - lo) * mt_double(s); }
/* numpy legacy random_interval (masked rejection, 32-bit path) */
static uint32_t mt_interval(mt19937* s, uint32_t max)
{
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (mt_next(s) & mask)) > max) {}
    return v;
}
/* gym.utils.seeding.np_random(seed): RandomState seeded with the uint32 words of
 * sha512(str(seed))[:8] (little endian), high zero words dropped */
static void gym_seed(mt19937* s, uint64_t seed)
{
    char txt[32];
    int n = snprintf(txt, sizeof(txt), "%llu", (unsigned long long)seed);
    uint8_t dig[64];
    sha512((const uint8_t*)txt, (size_t)n, dig);
    uint32_t w0 = (uint32_t)dig[0] | ((uint32_t)dig[1] << 8) | ((uint32_t)dig[2] << 16) | ((uint32_t)dig[3] << 24);
    uint32_t w1 = (uint32_t)dig[4] | ((uint32_t)dig[5] << 8) | ((uint32_t)dig[6] << 16) | ((uint32_t)dig[7] << 24);
    uint32_t key[2] = {w0, w1};
    int klen = w1 ? 2 : 1; /* _int_list_from_bigint drops high zeros; bigint 0 -> [0] */
    mt_init_by_array(s, key, klen);
}

void pmgo_rng_probe(uint64_t seed, int n_double, double* out_double, int shuffle_n, int32_t* out_perm)
{
    mt19937 s;
    gym_seed(&s, seed);
    for (int i = 0; i < n_double; i++) out_double[i] = mt_double(&s);
    for (int i = 0; i < shuffle_n; i++) out_perm[i] = i;
    for (int i = shuffle_n - 1; i >= 1; i--) {
        uint32_t j = mt_interval(&s, (uint32_t)i);
        int32_t t = out_perm[i]; out_perm[i] = out_perm[j]; out_perm[j] = t;
    }
}

/* ------------------------------------------------------------------ */
/* kinematics of the 17-link PyBullet tree                             */
/* ------------------------------------------------------------------ */
typedef struct {
    real R[NL][9];   /* world <- link frame */
    real p[NL][3];   /* link frame origin, world */
    real c[NL][3];   /* link COM, world */
    real S[NJ][6];   /* joint motion subspace, world coords about the world origin: [w; v_O] */
    real axis[NJ][3];
} Kin;

static void kinematics(const real* q, Kin* k)
{
    for (int i = 0; i < NL; i++) {
        int par = JPARENT_BL[i];
        real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
        if (par >= 0) { memcpy(Rp, k->R[par], sizeof(Rp)); v3cpy(pp, k->p[par]); }
        real Rj[9], xyz[3], ax[3];
        for (int a = 0; a < 3; a++) {
            xyz[a] = (real)BL_XYZ[i][a]; ax[a] = (real)BL_AXIS[i][a];
            for (int b = 0; b < 3; b++) Rj[3 * a + b] = (real)BL_ROT[i][a][b];
        }
        real R0[9];
        m3m(R0, Rp, Rj);
        real off[3];
        m3v(off, Rp, xyz);
        v3add(k->p[i], pp, off);
        int d = BL_DOF[i];
        if (BL_TYPE[i] == 0) { /* revolute about axis (here always local z) */
            real cq = RCOS(q[d]), sq = RSIN(q[d]);
            real x = ax[0], y = ax[1], z = ax[2], t = 1 - cq;
            real Rq[9] = {t * x * x + cq, t * x * y - sq * z, t * x * z + sq * y,
                          t * x * y + sq * z, t * y * y + cq, t * y * z - sq * x,
                          t * x * z - sq * y, t * y * z + sq * x, t * z * z + cq};
            m3m(k->R[i], R0, Rq);
        } else {
            memcpy(k->R[i], R0, sizeof(R0));
            if (BL_TYPE[i] == 1) {
                real aw[3];
                m3v(aw, R0, ax);
                v3axpy(k->p[i], q[d], aw);
            }
        }
        real com[3] = {(real)BL_COM[i][0], (real)BL_COM[i][1], (real)BL_COM[i][2]}, cw[3];
        m3v(cw, k->R[i], com);
        v3add(k->c[i], k->p[i], cw);
        if (d >= 0) {
            real aw[3];
            m3v(aw, k->R[i], ax);
            v3cpy(k->axis[d], aw);
            if (BL_TYPE[i] == 0) {
                v3cpy(k->S[d], aw);
                v3cross(k->S[d] + 3, k->p[i], aw); /* v_O = p x a */
            } else {
                v3set(k->S[d], 0, 0, 0);
                v3cpy(k->S[d] + 3, aw);
            }
        }
    }
}

/* is dof d an ancestor-or-self joint of Bullet link L? */
static int dof_affects_link(int d, int L)
{
    int i = L;
    while (i >= 0) {
        if (BL_DOF[i] == d) return 1;
        i = JPARENT_BL[i];
    }
    return 0;
}

/* row of the point Jacobian: (d/dq_d of the world velocity of point r fixed on link L) . n */
static void point_jacobian(const Kin* k, int L, const real* r, const real* n, real* J)
{
    for (int d = 0; d < NJ; d++) {
        J[d] = 0;
        if (!dof_affects_link(d, L)) continue;
        real v[3];
        v3cross(v, k->S[d], r); /* w x r */
        v3add(v, v, k->S[d] + 3);
        J[d] = v3dot(v, n);
    }
}
static void point_velocity(const Kin* k, const real* qd, int L, const real* r, real* vlin, real* w)
{
    v3set(vlin, 0, 0, 0);
    v3set(w, 0, 0, 0);
    for (int d = 0; d < NJ; d++) {
        if (!dof_affects_link(d, L)) continue;
        real v[3];
        v3cross(v, k->S[d], r);
        v3add(v, v, k->S[d] + 3);
        v3axpy(vlin, qd[d], v);
        v3axpy(w, qd[d], k->S[d]);
    }
}

void pmgo_fk_tip(const double q[9], double pos[3], double rot[9])
{
    real qq[NJ];
    for (int i = 0; i < NJ; i++) qq[i] = (real)q[i];
    Kin k;
    kinematics(qq, &k);
    for (int i = 0; i < 3; i++) pos[i] = k.p[PMG_BL_TIP][i];
    for (int i = 0; i < 9; i++) rot[i] = k.R[PMG_BL_TIP][i];
}

/* ------------------------------------------------------------------ */
/* articulated-body algorithm, world coordinates                       */
/* [BULLET-PRIOR] btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof */
/* ------------------------------------------------------------------ */
typedef struct {
    real U[NJ][6];
    real D[NJ];
} AbaCache;

static void sp_crm(real* o, const real* v, const real* m) /* motion x motion */
{
    real a[3], b[3], c[3];
    v3cross(a, v, m);
    v3cross(b, v, m + 3);
    v3cross(c, v + 3, m);
    v3cpy(o, a);
    v3add(o + 3, b, c);
}
static void sp_crf(real* o, const real* v, const real* f) /* motion x* force */
{
    real a[3], b[3], c[3];
    v3cross(a, v, f);
    v3cross(b, v + 3, f + 3);
    v3cross(c, v, f + 3);
    v3add(o, a, b);
    v3cpy(o + 3, c);
}
static void mat6v(real* o, const real* M, const real* v)
{
    real t[6];
    for (int i = 0; i < 6; i++) {
        real s = 0;
        for (int j = 0; j < 6; j++) s += M[6 * i + j] * v[j];
        t[i] = s;
    }
    memcpy(o, t, sizeof(t));
}
/* spatial inertia about the world origin of a body (mass m, COM c, rotational inertia Ic world) */
static void spatial_inertia(real* I, real m, const real* c, const real* Ic)
{
    real cx[9] = {0, -c[2], c[1], c[2], 0, -c[0], -c[1], c[0], 0};
    memset(I, 0, 36 * sizeof(real));
    real cc = v3dot(c, c);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            I[6 * a + b] = Ic[3 * a + b] + m * ((a == b ? cc : 0) - c[a] * c[b]);
            I[6 * a + 3 + b] = m * cx[3 * a + b];
            I[6 * (3 + a) + b] = m * cx[3 * b + a];
        }
    for (int a = 0; a < 3; a++) I[6 * (3 + a) + 3 + a] = m;
}

/* forward dynamics: qdd = M^-1 (tau - bias) with gravity + Bullet's per-link damping */
static void aba(const Kin* k, const real* qd, const real* tau, real* qdd, AbaCache* cache)
{
    static const real zero6[6] = {0, 0, 0, 0, 0, 0};
    real v[NL][6], cb[NL][6], IA[NL][36], pA[NL][6], u[NJ];
    for (int i = 0; i < NL; i++) {
        int par = JPARENT_BL[i], d = BL_DOF[i];
        memcpy(v[i], par >= 0 ? v[par] : zero6, sizeof(zero6));
        memset(cb[i], 0, sizeof(zero6));
        if (d >= 0) {
            real vj[6];
            for (int a = 0; a < 6; a++) vj[a] = k->S[d][a] * qd[d];
            for (int a = 0; a < 6; a++) v[i][a] += vj[a];
            sp_crm(cb[i], v[i], vj);
        }
        /* world rotational inertia about the COM */
        real Ic[9], tmp[9];
        const real* R = k->R[i];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) tmp[3 * a + b] = R[3 * a + b] * (real)BL_INERTIA[i][b];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                Ic[3 * a + b] = tmp[3 * a] * R[3 * b] + tmp[3 * a + 1] * R[3 * b + 1] + tmp[3 * a + 2] * R[3 * b + 2];
        real m = (real)BL_MASS[i];
        spatial_inertia(IA[i], m, k->c[i], Ic);
        real h[6];
        mat6v(h, IA[i], v[i]);
        sp_crf(pA[i], v[i], h);
        /* external: gravity at the COM + Bullet link damping (k1 = k2 = 0.04) */
        real vc[3], w[3];
        v3cpy(w, v[i]);
        v3cross(vc, w, k->c[i]);
        v3add(vc, vc, v[i] + 3);
        real f[3], n[3], Iw[3];
        real kl = LINK_DAMPING * (1 + v3norm(vc)), ka = LINK_DAMPING * (1 + v3norm(w));
        v3set(f, -m * vc[0] * kl, -m * vc[1] * kl, m * (-GRAVITY) - m * vc[2] * kl);
        m3v(Iw, Ic, w);
        v3set(n, -Iw[0] * ka, -Iw[1] * ka, -Iw[2] * ka);
        real cf[3];
        v3cross(cf, k->c[i], f);
        v3add(n, n, cf);
        for (int a = 0; a < 3; a++) { pA[i][a] -= n[a]; pA[i][3 + a] -= f[a]; }
    }
    for (int i = NL - 1; i >= 0; i--) {
        int par = JPARENT_BL[i], d = BL_DOF[i];
        real Ia[36], pa[6];
        if (d >= 0) {
            real* U = cache->U[d];
            mat6v(U, IA[i], k->S[d]);
            real D = 0, sp = 0;
            for (int a = 0; a < 6; a++) { D += k->S[d][a] * U[a]; sp += k->S[d][a] * pA[i][a]; }
            cache->D[d] = D;
            u[d] = tau[d] - sp;
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) Ia[6 * a + b] = IA[i][6 * a + b] - U[a] * U[b] / D;
            real Iac[6];
            mat6v(Iac, Ia, cb[i]);
            for (int a = 0; a < 6; a++) pa[a] = pA[i][a] + Iac[a] + U[a] * u[d] / D;
        } else {
            memcpy(Ia, IA[i], sizeof(Ia));
            memcpy(pa, pA[i], sizeof(pa));
        }
        if (par >= 0) {
            for (int a = 0; a < 36; a++) IA[par][a] += Ia[a];
            for (int a = 0; a < 6; a++) pA[par][a] += pa[a];
        }
    }
    real acc[NL][6];
    for (int i = 0; i < NL; i++) {
        int par = JPARENT_BL[i], d = BL_DOF[i];
        for (int a = 0; a < 6; a++) acc[i][a] = (par >= 0 ? acc[par][a] : 0) + cb[i][a];
        if (d >= 0) {
            real ua = 0;
            for (int a = 0; a < 6; a++) ua += cache->U[d][a] * acc[i][a];
            qdd[d] = (u[d] - ua) / cache->D[d];
            for (int a = 0; a < 6; a++) acc[i][a] += k->S[d][a] * qdd[d];
        }
    }
}

/* impulse response out = M^-1 f using the cached articulated quantities.
 * [BULLET-PRIOR] btMultiBody::calcAccelerationDeltasMultiDof */
static void aba_response(const Kin* k, const AbaCache* cache, const real* f, real* out)
{
    real pA[NL][6], u[NJ], acc[NL][6];
    memset(pA, 0, sizeof(pA));
    for (int i = NL - 1; i >= 0; i--) {
        int par = JPARENT_BL[i], d = BL_DOF[i];
        real pa[6];
        memcpy(pa, pA[i], sizeof(pa));
        if (d >= 0) {
            real sp = 0;
            for (int a = 0; a < 6; a++) sp += k->S[d][a] * pA[i][a];
            u[d] = f[d] - sp;
            for (int a = 0; a < 6; a++) pa[a] += cache->U[d][a] * u[d] / cache->D[d];
        }
        if (par >= 0)
            for (int a = 0; a < 6; a++) pA[par][a] += pa[a];
    }
    for (int i = 0; i < NL; i++) {
        int par = JPARENT_BL[i], d = BL_DOF[i];
        for (int a = 0; a < 6; a++) acc[i][a] = par >= 0 ? acc[par][a] : 0;
        if (d >= 0) {
            real ua = 0;
            for (int a = 0; a < 6; a++) ua += cache->U[d][a] * acc[i][a];
            out[d] = (u[d] - ua) / cache->D[d];
            for (int a = 0; a < 6; a++) acc[i][a] += k->S[d][a] * out[d];
        }
    }
}

void pmgo_fdyn(const double q[9], const double qd[9], const double tau[9], double qdd[9])
{
    real qq[NJ], vv[NJ], tt[NJ], aa[NJ];
    for (int i = 0; i < NJ; i++) { qq[i] = (real)q[i]; vv[i] = (real)qd[i]; tt[i] = (real)tau[i]; }
    Kin k;
    AbaCache c;
    kinematics(qq, &k);
    aba(&k, vv, tt, aa, &c);
    for (int i = 0; i < NJ; i++) qdd[i] = aa[i];
}
void pmgo_minv(const double q[9], double minv[81])
{
    real qq[NJ], z[NJ], aa[NJ];
    for (int i = 0; i < NJ; i++) { qq[i] = (real)q[i]; z[i] = 0; }
    Kin k;
    AbaCache c;
    kinematics(qq, &k);
    aba(&k, z, z, aa, &c);
    for (int j = 0; j < NJ; j++) {
        real f[NJ], o[NJ];
        for (int i = 0; i < NJ; i++) f[i] = (i == j);
        aba_response(&k, &c, f, o);
        for (int i = 0; i < NJ; i++) minv[9 * i + j] = o[i];
    }
}

/* ------------------------------------------------------------------ */
/* inverse kinematics                                                   */
/* [BULLET-PRIOR] PhysicsServerCommandProcessor::processCalculateInverseKinematics +
 * IKTrajectoryHelper::computeIK, method IK2_VEL_DLS_WITH_ORIENTATION (the 7-entry
 * null-space lists of kuka.py:272-277 do not match the 9 dofs, so pybullet.c drops
 * them), Jacobian::CalcDeltaThetasDLS2 with per-joint damping 0.5.           */
/* ------------------------------------------------------------------ */
static int solve_linear(int n, real* A, real* b) /* Gaussian elimination, partial pivoting; A n x n row-major */
{
    for (int c = 0; c < n; c++) {
        int piv = c;
        real best = RFABS(A[n * c + c]);
        for (int r = c + 1; r < n; r++)
            if (RFABS(A[n * r + c]) > best) { best = RFABS(A[n * r + c]); piv = r; }
        if (best == 0) return -1;
        if (piv != c) {
            for (int j = 0; j < n; j++) { real t = A[n * c + j]; A[n * c + j] = A[n * piv + j]; A[n * piv + j] = t; }
            real t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        for (int r = c + 1; r < n; r++) {
            real f = A[n * r + c] / A[n * c + c];
            for (int j = c; j < n; j++) A[n * r + j] -= f * A[n * c + j];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        real s = b[r];
        for (int j = r + 1; j < n; j++) s -= A[n * r + j] * b[j];
        b[r] = s / A[n * r + r];
    }
    return 0;
}

static int ik_solve(const real* q_start, const real* target, const real* tquat, int max_iter, real thr, real* q_out)
{
    real q[NJ];
    memcpy(q, q_start, sizeof(q));
    real diff = (real)1e30;
    int it = 0;
    for (; it < max_iter && diff > thr; it++) {
        Kin k;
        kinematics(q, &k);
        const real* p = k.p[PMG_BL_TIP];
        real e[6];
        v3sub(e, target, p);
        diff = v3norm(e);
        /* orientation error: deltaQ = endQ * startQ^-1, angle*axis */
        real sq[4], si[4], dq[4];
        R_to_quat(k.R[PMG_BL_TIP], sq);
        si[0] = -sq[0]; si[1] = -sq[1]; si[2] = -sq[2]; si[3] = sq[3];
        quat_mul(dq, tquat, si);
        real w = dq[3] < -1 ? -1 : (dq[3] > 1 ? 1 : dq[3]);
        real angle = 2 * RACOS(w);
        real s2 = 1 - dq[3] * dq[3];
        real ax[3] = {1, 0, 0};
        if (s2 >= (real)10.0 * (real)2.220446049250313e-16) {
            real s = 1 / RSQRT(s2);
            v3set(ax, dq[0] * s, dq[1] * s, dq[2] * s);
        }
        if (angle > PI_R) angle -= 2 * PI_R;
        real an = v3norm(ax);
        for (int a = 0; a < 3; a++) e[3 + a] = angle * ax[a] / an;
        /* Jacobian 6 x 9 of the tip link frame */
        real J[6][NJ];
        for (int d = 0; d < NJ; d++) {
            if (!dof_affects_link(d, PMG_BL_TIP)) {
                for (int r = 0; r < 6; r++) J[r][d] = 0;
                continue;
            }
            real v[3];
            v3cross(v, k.S[d], p);
            v3add(v, v, k.S[d] + 3);
            for (int a = 0; a < 3; a++) { J[a][d] = v[a]; J[3 + a][d] = k.S[d][a]; }
        }
        real U[NJ * NJ], rhs[NJ];
        for (int a = 0; a < NJ; a++) {
            for (int b = 0; b < NJ; b++) {
                real s = 0;
                for (int r = 0; r < 6; r++) s += J[r][a] * J[r][b];
                U[NJ * a + b] = s + (a == b ? IK_DAMPING : 0);
            }
            real s = 0;
            for (int r = 0; r < 6; r++) s += J[r][a] * e[r];
            rhs[a] = s;
        }
        solve_linear(NJ, U, rhs);
        real mx = 0;
        for (int a = 0; a < NJ; a++) mx = RFABS(rhs[a]) > mx ? RFABS(rhs[a]) : mx;
        if (mx > IK_MAX_STEP)
            for (int a = 0; a < NJ; a++) rhs[a] *= IK_MAX_STEP / mx;
        for (int a = 0; a < NJ; a++) q[a] += rhs[a];
    }
    memcpy(q_out, q, sizeof(q));
    return it;
}

int pmgo_ik(const double q_start[9], const double target_pos[3], const double target_quat_xyzw[4], int max_iter,
            double threshold, double q_out[9])
{
    real q[NJ], t[3], tq[4], o[NJ];
    for (int i = 0; i < NJ; i++) q[i] = (real)q_start[i];
    for (int i = 0; i < 3; i++) t[i] = (real)target_pos[i];
    for (int i = 0; i < 4; i++) tq[i] = (real)target_quat_xyzw[i];
    int it = ik_solve(q, t, tq, max_iter, (real)threshold, o);
    for (int i = 0; i < NJ; i++) q_out[i] = o[i];
    return it;
}

/* ------------------------------------------------------------------ */
/* box-box narrowphase: SAT over 15 axes + face clipping / edge-edge     */
/* [BULLET-PRIOR] btBoxBoxDetector (ODE dBoxBox2): same axis test, edge
 * fudge factor and <=4 contacts; the persistent manifold is replaced by
 * regeneration every substep with a CONTACT_MARGIN speculative distance. */
/* ------------------------------------------------------------------ */
typedef struct {
    real pa[3], pb[3]; /* world points on A and on B */
    real n[3];         /* unit normal pointing from B to A */
    real dist;         /* signed distance (negative = penetration) */
} CPoint;

static int clip_poly(const real (*in)[2], int n, real (*out)[2], int axis, real sign, real lim)
{
    /* keep sign*p[axis] <= lim */
    int m = 0;
    for (int i = 0; i < n; i++) {
        const real* a = in[i];
        const real* b = in[(i + 1) % n];
        real da = sign * a[axis] - lim, db = sign * b[axis] - lim;
        if (da <= 0) { out[m][0] = a[0]; out[m][1] = a[1]; m++; }
        if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
            real t = da / (da - db);
            out[m][0] = a[0] + t * (b[0] - a[0]);
            out[m][1] = a[1] + t * (b[1] - a[1]);
            m++;
        }
    }
    return m;
}

static int box_box(const real* ca, const real* Ra, const real* ha, const real* cb, const real* Rb, const real* hb,
                   real margin, CPoint* out)
{
    real A[3][3], B[3][3];
    for (int i = 0; i < 3; i++)
        for (int a = 0; a < 3; a++) { A[i][a] = Ra[3 * a + i]; B[i][a] = Rb[3 * a + i]; }
    real d[3];
    v3sub(d, cb, ca);
    real C[3][3], Q[3][3], dA[3], dB[3];
    for (int i = 0; i < 3; i++) {
        dA[i] = v3dot(d, A[i]);
        dB[i] = v3dot(d, B[i]);
        for (int j = 0; j < 3; j++) { C[i][j] = v3dot(A[i], B[j]); Q[i][j] = RFABS(C[i][j]); }
    }
    real best = (real)-1e30;
    int code = -1;
    real nrm[3] = {0, 0, 0}; /* axis direction from A towards B */
    for (int i = 0; i < 3; i++) {
        real s = RFABS(dA[i]) - (ha[i] + hb[0] * Q[i][0] + hb[1] * Q[i][1] + hb[2] * Q[i][2]);
        if (s > margin) return 0;
        if (s > best) { best = s; code = i; real sg = dA[i] < 0 ? (real)-1 : (real)1; v3set(nrm, sg * A[i][0], sg * A[i][1], sg * A[i][2]); }
    }
    for (int j = 0; j < 3; j++) {
        real s = RFABS(dB[j]) - (hb[j] + ha[0] * Q[0][j] + ha[1] * Q[1][j] + ha[2] * Q[2][j]);
        if (s > margin) return 0;
        if (s > best) { best = s; code = 3 + j; real sg = dB[j] < 0 ? (real)-1 : (real)1; v3set(nrm, sg * B[j][0], sg * B[j][1], sg * B[j][2]); }
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            real L[3];
            v3cross(L, A[i], B[j]);
            real len = v3norm(L);
            if (len < (real)1e-6) continue;
            real proj = v3dot(d, L);
            real ra = ha[i1] * Q[i2][j] + ha[i2] * Q[i1][j];
            real rb = hb[j1] * Q[i][j2] + hb[j2] * Q[i][j1];
            real s = (RFABS(proj) - (ra + rb)) / len;
            if (s > margin) return 0;
            /* edges must beat the best face by the fudge factor (penetrations scaled up, gaps scaled down) */
            real pen = s < 0 ? s * EDGE_FUDGE : s / EDGE_FUDGE;
            if (pen > best) {
                best = s; code = 6 + 3 * i + j;
                real sg = proj < 0 ? (real)-1 : (real)1;
                v3set(nrm, sg * L[0] / len, sg * L[1] / len, sg * L[2] / len);
            }
        }
    if (code < 0) return 0;
    if (code >= 6) {
        int i = (code - 6) / 3, j = (code - 6) % 3;
        real pa[3], pb[3];
        v3cpy(pa, ca);
        v3cpy(pb, cb);
        for (int kx = 0; kx < 3; kx++) {
            if (kx != i) { real sg = v3dot(nrm, A[kx]) > 0 ? (real)1 : (real)-1; v3axpy(pa, sg * ha[kx], A[kx]); }
            if (kx != j) { real sg = v3dot(nrm, B[kx]) > 0 ? (real)-1 : (real)1; v3axpy(pb, sg * hb[kx], B[kx]); }
        }
        /* closest points of lines pa + s A[i], pb + t B[j] */
        real r[3];
        v3sub(r, pb, pa);
        real uaub = C[i][j], q1 = v3dot(A[i], r), q2 = -v3dot(B[j], r);
        real den = 1 - uaub * uaub;
        real s = 0, t = 0;
        if (den > (real)1e-8) { s = (q1 + uaub * q2) / den; t = (uaub * q1 + q2) / den; }
        v3axpy(pa, s, A[i]);
        v3axpy(pb, t, B[j]);
        v3cpy(out[0].pa, pa);
        v3cpy(out[0].pb, pb);
        v3set(out[0].n, -nrm[0], -nrm[1], -nrm[2]);
        out[0].dist = best;
        return 1;
    }
    /* face contact: reference box owns the axis */
    int refA = code < 3;
    const real *cr = refA ? ca : cb, *ci = refA ? cb : ca, *hr = refA ? ha : hb, *hi = refA ? hb : ha;
    real (*Rr)[3] = refA ? A : B;
    real (*Ri)[3] = refA ? B : A;
    int ax = refA ? code : code - 3;
    real nr[3]; /* outward reference-face normal, pointing at the incident box */
    if (refA) v3cpy(nr, nrm); else v3set(nr, -nrm[0], -nrm[1], -nrm[2]);
    /* incident face: most anti-parallel axis */
    int ia = 0;
    real bestd = -1;
    for (int kx = 0; kx < 3; kx++) {
        real dd = RFABS(v3dot(nr, Ri[kx]));
        if (dd > bestd) { bestd = dd; ia = kx; }
    }
    real isg = v3dot(nr, Ri[ia]) > 0 ? (real)-1 : (real)1;
    int iu = (ia + 1) % 3, iv = (ia + 2) % 3, ru = (ax + 1) % 3, rv = (ax + 2) % 3;
    real fc[3];
    v3cpy(fc, ci);
    v3axpy(fc, isg * hi[ia], Ri[ia]);
    real poly[8][2], tmp[8][2];
    static const real SU[4] = {1, -1, -1, 1}, SV[4] = {1, 1, -1, -1};
    real vz[4];
    for (int c = 0; c < 4; c++) {
        real v[3], rel[3];
        v3cpy(v, fc);
        v3axpy(v, SU[c] * hi[iu], Ri[iu]);
        v3axpy(v, SV[c] * hi[iv], Ri[iv]);
        v3sub(rel, v, cr);
        poly[c][0] = v3dot(rel, Rr[ru]);
        poly[c][1] = v3dot(rel, Rr[rv]);
        vz[c] = v3dot(rel, nr);
    }
    /* plane of the incident face in reference coords: z = z0 + gu*u + gv*v */
    real e1u = poly[1][0] - poly[0][0], e1v = poly[1][1] - poly[0][1], e1z = vz[1] - vz[0];
    real e2u = poly[3][0] - poly[0][0], e2v = poly[3][1] - poly[0][1], e2z = vz[3] - vz[0];
    real det = e1u * e2v - e1v * e2u;
    real gu = 0, gv = 0;
    if (RFABS(det) > (real)1e-12) { gu = (e1z * e2v - e2z * e1v) / det; gv = (e2z * e1u - e1z * e2u) / det; }
    real z0 = vz[0] - gu * poly[0][0] - gv * poly[0][1];
    int n = 4;
    n = clip_poly(poly, n, tmp, 0, 1, hr[ru]);
    n = clip_poly(tmp, n, poly, 0, -1, hr[ru]);
    n = clip_poly(poly, n, tmp, 1, 1, hr[rv]);
    n = clip_poly(tmp, n, poly, 1, -1, hr[rv]);
    real pts[8][3], sep[8];
    int m = 0;
    for (int c = 0; c < n; c++) {
        real z = z0 + gu * poly[c][0] + gv * poly[c][1];
        real s = z - hr[ax];
        if (s > margin) continue;
        pts[m][0] = poly[c][0]; pts[m][1] = poly[c][1]; pts[m][2] = z;
        sep[m] = s;
        m++;
    }
    if (m == 0) return 0;
    int sel[4], ns = 0;
    if (m <= 4) {
        for (int c = 0; c < m; c++) sel[ns++] = c;
    } else {
        /* deepest, farthest from it, then largest triangle on either side */
        int i0 = 0;
        for (int c = 1; c < m; c++) if (sep[c] < sep[i0]) i0 = c;
        int i1 = -1; real bd = -1;
        for (int c = 0; c < m; c++) {
            if (c == i0) continue;
            real du = pts[c][0] - pts[i0][0], dv = pts[c][1] - pts[i0][1];
            real dd = du * du + dv * dv;
            if (dd > bd) { bd = dd; i1 = c; }
        }
        int i2 = -1, i3 = -1; real amax = 0, amin = 0;
        for (int c = 0; c < m; c++) {
            if (c == i0 || c == i1) continue;
            real ar = (pts[i1][0] - pts[i0][0]) * (pts[c][1] - pts[i0][1]) - (pts[i1][1] - pts[i0][1]) * (pts[c][0] - pts[i0][0]);
            if (ar > amax) { amax = ar; i2 = c; }
            if (ar < amin) { amin = ar; i3 = c; }
        }
        sel[ns++] = i0; sel[ns++] = i1;
        if (i2 >= 0) sel[ns++] = i2;
        if (i3 >= 0) sel[ns++] = i3;
    }
    for (int c = 0; c < ns; c++) {
        int s = sel[c];
        real pin[3], pref[3];
        v3cpy(pin, cr);
        v3axpy(pin, pts[s][0], Rr[ru]);
        v3axpy(pin, pts[s][1], Rr[rv]);
        v3cpy(pref, pin);
        v3axpy(pin, pts[s][2], nr);
        v3axpy(pref, hr[ax], nr);
        if (refA) { v3cpy(out[c].pa, pref); v3cpy(out[c].pb, pin); v3set(out[c].n, -nr[0], -nr[1], -nr[2]); }
        else { v3cpy(out[c].pa, pin); v3cpy(out[c].pb, pref); v3cpy(out[c].n, nr); }
        out[c].dist = sep[s];
    }
    return ns;
}

/* ------------------------------------------------------------------ */
/* cylinder (A) x box (B) narrowphase.  Bullet runs GJK/EPA (btConvexConvexAlgorithm) and lets the
 * persistent manifold collect up to 4 points over frames; this restatement regenerates a <= 4 point
 * manifold every substep from a finite separating-axis search (3 box face normals, the cylinder
 * axis, 3 axis x edge directions, 1 closest-feature axis) and feature clipping:
 *   box face + cap parallel  -> rim points inside the face and face corners inside the disc
 *   box face + side          -> the two ends of the deepest generator, clamped to the face
 *   cylinder axis            -> box vertices on the supporting feature that lie inside the disc
 *   edge / closest-feature   -> one point from the closest points of axis segment and box
 * (build choice).  n points from the box to the cylinder.                */
static real box_proj(const real (*B)[3], const real* hb, const real* L)
{
    return hb[0] * RFABS(v3dot(B[0], L)) + hb[1] * RFABS(v3dot(B[1], L)) + hb[2] * RFABS(v3dot(B[2], L));
}
static void closest_on_box(const real* cb, const real (*B)[3], const real* hb, const real* p, real* q)
{
    real d[3];
    v3sub(d, p, cb);
    v3cpy(q, cb);
    for (int k = 0; k < 3; k++) {
        real t = v3dot(d, B[k]);
        t = t < -hb[k] ? -hb[k] : (t > hb[k] ? hb[k] : t);
        v3axpy(q, t, B[k]);
    }
}
/* closest point of the solid cylinder (centre cc, axis a, radius rad, half length hl) to p */
static void closest_on_cyl(const real* cc, const real* a, real rad, real hl, const real* p, real* q)
{
    real w[3], r[3];
    v3sub(w, p, cc);
    real wa = v3dot(w, a);
    real ta = wa < -hl ? -hl : (wa > hl ? hl : wa);
    for (int x = 0; x < 3; x++) r[x] = w[x] - wa * a[x];
    real rl = v3norm(r);
    real sc = rl > rad ? rad / rl : 1;
    for (int x = 0; x < 3; x++) q[x] = cc[x] + ta * a[x] + sc * r[x];
}
/* The deepest point of a box FACE under a tilted cap that is neither a face corner inside the disc nor a rim point over
 * the face: where an edge of the face crosses the cap's rim (seen along the cylinder axis).  The height above the cap
 * plane is linear over the face, so its extreme over face x disc sits at a corner, at the lowest rim point or at such a
 * crossing -- the candidate the clipped feature points lacked (a contact found up to 1.7 mm late at <= 3 degrees of tilt,
 * tests/test_oracle_physics.py).  fc: face centre, (B1, h1), (B2, h2): its in-plane axes; pc: cap centre, a: cylinder
 * axis; the contact normal n, can = a . n.  Of the <= 8 crossings the one with the smallest separation t along n (box
 * point q, cylinder point q + t n in the cap plane) is returned; 0 if no edge crosses the rim. */
static int face_rim_crossing(const real* fc, const real* B1, real h1, const real* B2, real h2, const real* pc, const real* a,
                             real rad, const real* n, real can, real* q, real* tq)
{
    int found = 0;
    real bt = 0;
    for (int e = 0; e < 4; e++) {
        const real* E = e < 2 ? B2 : B1;                 /* edge direction; the edge sits at +-h of the OTHER axis */
        const real* F = e < 2 ? B1 : B2;
        real he = e < 2 ? h2 : h1, hf = (e & 1) ? (e < 2 ? h1 : h2) : -(e < 2 ? h1 : h2);
        real w0[3];
        for (int x = 0; x < 3; x++) w0[x] = fc[x] + hf * F[x] - pc[x];
        real Ea = v3dot(E, a), wa = v3dot(w0, a);
        real qa = 1 - Ea * Ea, qb = v3dot(w0, E) - wa * Ea, qc = v3dot(w0, w0) - wa * wa - rad * rad;
        real disc = qb * qb - qa * qc;
        if (qa < (real)1e-9 || disc < 0) continue;
        real sq = RSQRT(disc);
        for (int r2 = 0; r2 < 2; r2++) {
            real t = (-qb + (r2 ? sq : -sq)) / qa;
            if (RFABS(t) > he) continue;
            real pt[3];
            for (int x = 0; x < 3; x++) pt[x] = w0[x] + t * E[x];      /* relative to the cap centre */
            real ts = -v3dot(pt, a) / can;
            if (!found || ts < bt) { found = 1; bt = ts; for (int x = 0; x < 3; x++) q[x] = pt[x] + pc[x]; }
        }
    }
    *tq = bt;
    return found;
}
/* mutual closest points of the cylinder and the box by alternating projections (disjoint convex sets) */
static void closest_rounds(const real* cc, const real* a, real rad, real hl, const real* cb, const real (*B)[3],
                           const real* hb, real* qc, real* p0)
{
    /* six rounds of alternating projections (a crawl of 0.3 % per round when the two nearest features are almost parallel:
     * what follows, and the second start of cyl_box_closest, deal with the case that matters, a cap over a box edge.
     * Extrapolating the crawl to its limit was tried: it doubles the float32 / float64 divergence of the chest tasks'
     * handle contacts, where the minimum is flat, for a few 1e-5 of accuracy here) */
    for (int it = 0; it < 6; it++) {
        closest_on_cyl(cc, a, rad, hl, p0, qc);
        closest_on_box(cb, B, hb, qc, p0);
    }
    /* (qc, p0 = the box's closest point to qc: their difference is a normal of the box at p0 -- the supporting direction
     * the axis test wants -- whether the rounds have settled or not) */
    {   /* p0 on an EDGE of the box (two coordinates at their extents): the rim of a tilted cap against that edge is the slowest
         * crawl of all, and a one-dimensional problem -- the squared distance to the cylinder is convex along the edge and
         * its derivative is (x - closest(x)) . E: eight regula-falsi (Illinois) steps on it */
        real c[3], w[3];
        v3sub(w, p0, cb);
        int nfix = 0, kf = -1;
        for (int k = 0; k < 3; k++) {
            c[k] = v3dot(w, B[k]);
            if (RFABS(c[k]) >= hb[k] * (1 - (real)1e-6)) nfix++; else kf = k;
        }
        real wq[3];
        v3sub(wq, qc, cc);
        if (nfix == 2 && RFABS(v3dot(wq, a)) >= hl * (1 - (real)1e-6)) {      /* ... and the cylinder's point on a cap */
            real base[3], x[3], q[3];
            v3cpy(base, p0); v3axpy(base, -c[kf], B[kf]);
            real t0 = -hb[kf], t1 = hb[kf], g0, g1;
            v3cpy(x, base); v3axpy(x, t0, B[kf]); closest_on_cyl(cc, a, rad, hl, x, q); v3sub(w, x, q); g0 = v3dot(w, B[kf]);
            v3cpy(x, base); v3axpy(x, t1, B[kf]); closest_on_cyl(cc, a, rad, hl, x, q); v3sub(w, x, q); g1 = v3dot(w, B[kf]);
            real t = c[kf];
            if (g0 < 0 && g1 > 0) {
                for (int it = 0; it < 8; it++) {
                    t = (t0 * g1 - t1 * g0) / (g1 - g0);
                    v3cpy(x, base); v3axpy(x, t, B[kf]); closest_on_cyl(cc, a, rad, hl, x, q); v3sub(w, x, q);
                    real g = v3dot(w, B[kf]);
                    if (g > 0) { t1 = t; g1 = g; g0 *= (real)0.5; } else { t0 = t; g0 = g; g1 *= (real)0.5; }
                }
                v3cpy(x, base); v3axpy(x, t, B[kf]);
                closest_on_cyl(cc, a, rad, hl, x, q);
                v3sub(w, q, x);
                real e0[3];
                v3sub(e0, qc, p0);
                if (v3dot(w, w) < (real)0.999 * v3dot(e0, e0)) { v3cpy(qc, q); closest_on_box(cb, B, hb, qc, p0); }
            }
        }
    }
}
/* Two starts: the box's closest point to the cylinder's centre, and -- when a cap is within 8 degrees of parallel to a face
 * of the box -- the point where an edge of that face crosses the rim (face_rim_crossing): from the first one the rounds
 * crawl ACROSS the face towards that edge at 0.3 % per round.  The closer pair wins. */
static void cyl_box_closest(const real* cc, const real* a, real rad, real hl, const real* cb, const real (*B)[3],
                            const real* hb, real* qc, real* p0)
{
    closest_on_box(cb, B, hb, cc, p0);
    closest_rounds(cc, a, rad, hl, cb, B, hb, qc, p0);
    int kf = 0;
    real bf = RFABS(v3dot(B[0], a));
    for (int k = 1; k < 3; k++) { real x = RFABS(v3dot(B[k], a)); if (x > bf) { bf = x; kf = k; } }
    real wq0[3];
    v3sub(wq0, qc, cc);
    /* (a CAP against the box only: lateral contacts settle from the first start) */
    if (bf > (real)0.99 && bf < (real)0.9999995 && RFABS(v3dot(wq0, a)) >= hl * (1 - (real)1e-6)) {
        real d[3], n[3], pc[3], fc[3], q[3], tq;
        v3sub(d, cc, cb);
        real sg = v3dot(d, a) < 0 ? (real)-1 : (real)1;           /* n = +-a from the box towards the cylinder */
        v3set(n, sg * a[0], sg * a[1], sg * a[2]);
        v3cpy(pc, cc); v3axpy(pc, -hl, n);
        real nb = v3dot(n, B[kf]) > 0 ? (real)1 : (real)-1;
        v3cpy(fc, cb); v3axpy(fc, nb * hb[kf], B[kf]);
        int f1 = (kf + 1) % 3, f2 = (kf + 2) % 3;
        if (face_rim_crossing(fc, B[f1], hb[f1], B[f2], hb[f2], pc, n, rad, n, (real)1, q, &tq)) {
            real q2[3], e0[3], e1[3];
            closest_rounds(cc, a, rad, hl, cb, B, hb, q2, q);
            v3sub(e0, qc, p0); v3sub(e1, q2, q);
            if (v3dot(e1, e1) < (real)0.999 * v3dot(e0, e0)) { v3cpy(qc, q2); v3cpy(p0, q); }
        }
    }
}
static int reduce4(const real (*pts)[3], const real* sep, int m, int* sel)
{
    if (m <= 4) { for (int c = 0; c < m; c++) sel[c] = c; return m; }
    int i0 = 0;
    for (int c = 1; c < m; c++) if (sep[c] < sep[i0]) i0 = c;
    int i1 = -1; real bd = -1;
    for (int c = 0; c < m; c++) {
        if (c == i0) continue;
        real d[3]; v3sub(d, pts[c], pts[i0]);
        real dd = v3dot(d, d);
        if (dd > bd) { bd = dd; i1 = c; }
    }
    int i2 = -1, i3 = -1; real amax = 0, amin = 0;
    real e[3]; v3sub(e, pts[i1], pts[i0]);
    real ref[3] = {0, 0, 0};
    for (int c = 0; c < m; c++) {
        if (c == i0 || c == i1) continue;
        real f[3], x[3]; v3sub(f, pts[c], pts[i0]); v3cross(x, e, f);
        if (v3dot(ref, ref) == 0 && v3dot(x, x) > 0) v3cpy(ref, x);
        real ar = v3dot(x, ref);
        if (ar > amax) { amax = ar; i2 = c; }
        if (ar < amin) { amin = ar; i3 = c; }
    }
    int ns = 0;
    sel[ns++] = i0; sel[ns++] = i1;
    if (i2 >= 0) sel[ns++] = i2;
    if (i3 >= 0) sel[ns++] = i3;
    return ns;
}
static int cyl_box(const real* cc, const real* Rc, real rad, real hl, const real* cb, const real* Rb, const real* hb,
                   real margin, CPoint* out)
{
    real a[3] = {Rc[2], Rc[5], Rc[8]}, u[3] = {Rc[0], Rc[3], Rc[6]}, v[3] = {Rc[1], Rc[4], Rc[7]};
    real B[3][3];
    for (int k = 0; k < 3; k++) for (int x = 0; x < 3; x++) B[k][x] = Rb[3 * x + k];
    real d[3];
    v3sub(d, cc, cb);
    real best = (real)-1e30, bn[3] = {0, 0, 0};
    int btype = -1, bk = 0;
    real q3[3] = {0, 0, 0}, L3[3] = {0, 0, 0}, len3 = -1;   /* mutual closest pair (cylinder point, direction, distance) */
    /* type 0: box faces, 1: cylinder axis, 2: axis x edge, 3: closest feature */
    for (int pass = 0; pass < 8; pass++) {
        real L[3];
        int type, k = 0;
        if (pass < 3) { type = 0; k = pass; v3cpy(L, B[k]); }
        else if (pass == 3) { type = 1; v3cpy(L, a); }
        else if (pass < 7) {
            type = 2; k = pass - 4;
            v3cross(L, a, B[k]);
            real len = v3norm(L);
            if (len < (real)1e-6) continue;
            for (int x = 0; x < 3; x++) L[x] /= len;
        } else {
            type = 3; /* the direction between the mutual closest points (vertex / rim configurations) */
            real p0[3], qc[3];
            cyl_box_closest(cc, a, rad, hl, cb, (const real (*)[3])B, hb, qc, p0);
            v3sub(L, qc, p0);
            real len = v3norm(L);
            if (len < (real)1e-6) {        /* (1 um: what float32 resolves of two coinciding points at 0.5 m) */
                /* PENETRATING shapes have no closest pair.  What this pass finds for a box edge that runs ALONG the cylinder
                 * (|a . B_k| > 0.7: a finger's vertical edge against the puck's side) while the two are apart -- the radial
                 * direction through that edge -- is then taken from the geometry: the box vertex nearest to the cylinder's
                 * centre, seen across the axis.  Without it the least penetration was taken over the face normals alone: a
                 * finger corner 0.15 mm inside the puck's side was reported 8.5 mm deep along the finger's face normal, and
                 * the contact's error reduction launched the puck at 0.65 m/s (tools/strike_puck.py).  Treated as an
                 * axis x edge direction (one lateral contact point). */
                real bp = RFABS(v3dot(B[0], a));
                for (int q = 1; q < 3; q++) { real x = RFABS(v3dot(B[q], a)); if (x > bp) bp = x; }
                if (bp <= (real)0.7) continue;
                real c[3], dm[3];
                v3sub(dm, cb, cc);
                v3cpy(c, dm);
                for (int q = 0; q < 3; q++) v3axpy(c, v3dot(dm, B[q]) > 0 ? -hb[q] : hb[q], B[q]);
                v3axpy(c, -v3dot(c, a), a);
                real lc = v3norm(c);
                if (lc < (real)1e-6) continue;
                type = 2; k = 0;
                for (int x = 0; x < 3; x++) L[x] = c[x] / lc;
            } else {
                for (int x = 0; x < 3; x++) L[x] /= len;
                v3cpy(q3, qc); v3cpy(L3, L); len3 = len;
            }
        }
        real t = v3dot(d, L), ca = v3dot(a, L);
        real rc = hl * RFABS(ca) + rad * RSQRT(1 - ca * ca > 0 ? 1 - ca * ca : 0);
        real sep = RFABS(t) - (box_proj(B, hb, L) + rc);
        if (sep > margin) return 0;
        real pen = type >= 2 ? (sep < 0 ? sep * EDGE_FUDGE : sep / EDGE_FUDGE) : sep;
        if (pen > best) {
            best = sep; btype = type; bk = k;
            real sg = t < 0 ? (real)-1 : (real)1;
            v3set(bn, sg * L[0], sg * L[1], sg * L[2]);
        }
    }
    if (btype < 0) return 0;
    real pts[12][3], sep[12];   /* candidate points ON THE CYLINDER side and their signed distances */
    int m = 0;
    const real* n = bn;
    real can = v3dot(a, n);
    if (btype == 0) {
        real fp[3];
        v3cpy(fp, cb);
        v3axpy(fp, box_proj(B, hb, n), n);            /* a point of the supporting box face plane */
        int j1 = (bk + 1) % 3, j2 = (bk + 2) % 3;
        if (RFABS(can) >= (real)0.7) {
            real sg = can > 0 ? (real)-1 : (real)1;    /* the cap that faces the box */
            real pc[3];
            v3cpy(pc, cc); v3axpy(pc, sg * hl, a);
            {   /* a tilted cap touches with ONE rim point, the lowest along n: it leads the candidates (the four fixed
                 * samples below would only find the contact several mm too late) -- when it lies over the face.  When the
                 * cap hangs over an edge of the face the deepest point is where that edge crosses the rim */
                real md[3];
                for (int x = 0; x < 3; x++) md[x] = n[x] - can * a[x];
                real ml = v3norm(md);
                if (ml > (real)1e-3) {
                    real p[3], w[3];
                    v3cpy(p, pc); v3axpy(p, -rad / ml, md);
                    v3sub(w, p, cb);
                    if (RFABS(v3dot(w, B[j1])) <= hb[j1] && RFABS(v3dot(w, B[j2])) <= hb[j2]) {
                        v3sub(w, p, fp);
                        v3cpy(pts[m], p); sep[m] = v3dot(w, n); m++;
                    } else {
                        real fc[3], q[3], tq;
                        real nb = v3dot(n, B[bk]) > 0 ? (real)1 : (real)-1;
                        v3cpy(fc, cb); v3axpy(fc, nb * hb[bk], B[bk]);
                        if (face_rim_crossing(fc, B[j1], hb[j1], B[j2], hb[j2], pc, a, rad, n, can, q, &tq)) {
                            v3cpy(pts[m], q); v3axpy(pts[m], tq, n); sep[m] = tq; m++;
                        }
                    }
                }
            }
            for (int c = 0; c < 4; c++) {              /* rim points inside the face rectangle */
                real p[3], w[3];
                v3cpy(p, pc);
                v3axpy(p, (c == 0 ? rad : (c == 1 ? -rad : 0)), u);
                v3axpy(p, (c == 2 ? rad : (c == 3 ? -rad : 0)), v);
                v3sub(w, p, cb);
                if (RFABS(v3dot(w, B[j1])) <= hb[j1] && RFABS(v3dot(w, B[j2])) <= hb[j2]) {
                    v3cpy(pts[m], p); v3sub(w, p, fp); sep[m] = v3dot(w, n); m++;
                }
            }
            for (int c = 0; c < 4; c++) {              /* face corners inside the disc */
                real q[3], w[3];
                v3cpy(q, fp);
                real nb = v3dot(n, B[bk]) > 0 ? (real)1 : (real)-1;
                v3cpy(q, cb); v3axpy(q, nb * hb[bk], B[bk]);
                v3axpy(q, ((c & 1) ? hb[j1] : -hb[j1]), B[j1]);
                v3axpy(q, ((c & 2) ? hb[j2] : -hb[j2]), B[j2]);
                v3sub(w, q, pc);
                real wa = v3dot(w, a);
                if (v3dot(w, w) - wa * wa <= rad * rad) {
                    real t = -wa / can;                /* q + n t lies in the cap plane */
                    v3cpy(pts[m], q); v3axpy(pts[m], t, n); sep[m] = t; m++;
                }
            }
            if (m == 0) {                              /* partial overlap without a rim point / corner inside */
                real q[3], w[3];
                closest_on_box(cb, B, hb, pc, q);
                v3sub(w, q, pc);
                real wa = v3dot(w, a);
                v3axpy(w, -wa, a);
                real rho = v3norm(w);
                real p[3];
                v3cpy(p, pc);
                if (rho > (real)1e-9) v3axpy(p, (rho < rad ? rho : rad) / rho, w);
                v3sub(w, p, fp);
                v3cpy(pts[m], p); sep[m] = v3dot(w, n); m++;
            }
        } else {
            real mdir[3] = {n[0] - can * a[0], n[1] - can * a[1], n[2] - can * a[2]};
            real ml = v3norm(mdir);
            for (int x = 0; x < 3; x++) mdir[x] /= ml;
            for (int e2 = 0; e2 < 2; e2++) {           /* both ends of the generator closest to the face */
                real p[3], w[3];
                v3cpy(p, cc);
                v3axpy(p, -rad, mdir);
                v3axpy(p, e2 == 0 ? hl : -hl, a);
                v3sub(w, p, fp);
                real s = v3dot(w, n);
                if (s > margin) continue;
                v3sub(w, p, cb);                       /* clamp onto the face rectangle along the in-plane axes */
                real c1 = v3dot(w, B[j1]), c2 = v3dot(w, B[j2]);
                real k1 = c1 < -hb[j1] ? -hb[j1] : (c1 > hb[j1] ? hb[j1] : c1);
                real k2 = c2 < -hb[j2] ? -hb[j2] : (c2 > hb[j2] ? hb[j2] : c2);
                v3axpy(p, k1 - c1, B[j1]);
                v3axpy(p, k2 - c2, B[j2]);
                v3cpy(pts[m], p); sep[m] = s; m++;
            }
        }
    } else if (btype == 1) {
        real pc[3];
        v3cpy(pc, cc); v3axpy(pc, -hl, n);             /* cap facing the box (n = +-a) */
        real smax = (real)-1e30, sv[8], q8[8][3];
        for (int c = 0; c < 8; c++) {
            v3cpy(q8[c], cb);
            for (int k = 0; k < 3; k++) v3axpy(q8[c], ((c >> k) & 1) ? hb[k] : -hb[k], B[k]);
            real w[3];
            v3sub(w, q8[c], cb);
            sv[c] = v3dot(w, n);
            if (sv[c] > smax) smax = sv[c];
        }
        for (int c = 0; c < 8 && m < 8; c++) {
            if (sv[c] < smax - (real)1e-3) continue;
            real w[3];
            v3sub(w, q8[c], pc);
            real wa = v3dot(w, n);
            real rho2 = v3dot(w, w) - wa * wa;
            if (rho2 > rad * rad) continue;
            v3cpy(pts[m], q8[c]); v3axpy(pts[m], -wa, n); sep[m] = -wa; m++;
        }
        {   /* the box face that looks at the cap: where its edges cross the rim (face_rim_crossing) */
            int kf = 0;
            real bf = RFABS(v3dot(B[0], n));
            for (int k = 1; k < 3; k++) { real x = RFABS(v3dot(B[k], n)); if (x > bf) { bf = x; kf = k; } }
            int f1 = (kf + 1) % 3, f2 = (kf + 2) % 3;
            real fc[3], q[3], tq;
            real nb = v3dot(n, B[kf]) > 0 ? (real)1 : (real)-1;
            v3cpy(fc, cb); v3axpy(fc, nb * hb[kf], B[kf]);
            if (m < 8 && bf > (real)0.7 && bf < (real)0.9999995 &&
                face_rim_crossing(fc, B[f1], hb[f1], B[f2], hb[f2], pc, n, rad, n, (real)1, q, &tq)) {
                v3cpy(pts[m], q); v3axpy(pts[m], tq, n); sep[m] = tq; m++;
            }
        }
        if (m == 0) {
            real q[3], w[3];
            closest_on_box(cb, B, hb, pc, q);
            v3sub(w, q, pc);
            real wa = v3dot(w, n);
            v3cpy(pts[m], q); v3axpy(pts[m], -wa, n); sep[m] = -wa; m++;
        }
    } else {
        /* one point.  Closest-feature direction: the cylinder's point of the mutual closest pair.  Axis x edge direction
         * (n perpendicular to the axis, a lateral contact): the generator facing the box, at the axial position next to
         * the box found by alternating closest points of the axis segment and the box */
        if (btype == 3) {
            real p0[3];
            cyl_box_closest(cc, a, rad, hl, cb, (const real (*)[3])B, hb, pts[0], p0);
        } else {
            real p0[3], s0[3], w[3];
            closest_on_box(cb, B, hb, cc, p0);
            for (int it = 0; it < 4; it++) {
                v3sub(w, p0, cc);
                real t = v3dot(w, a);
                t = t < -hl ? -hl : (t > hl ? hl : t);
                v3cpy(s0, cc); v3axpy(s0, t, a);
                closest_on_box(cb, B, hb, s0, p0);
            }
            v3cpy(pts[0], s0); v3axpy(pts[0], -rad, n);
        }
        sep[0] = best;
        m = 1;
    }
    {   /* drop candidates beyond the speculative margin (keep the order) */
        int k2 = 0;
        for (int c = 0; c < m; c++)
            if (sep[c] <= margin) { if (k2 != c) { v3cpy(pts[k2], pts[c]); sep[k2] = sep[c]; } k2++; }
        m = k2;
        if (m == 0) {
            /* the face / axis case found overlap along its axis but no feature point within the margin: an edge or vertex
             * passing beside the rim.  The mutual closest pair IS the contact then; the alternating projections are
             * continued until they settle (an edge nearly parallel to the cap converges slowly, the six iterations of
             * the axis pass are not enough there) */
            if (len3 < 0) return 0;
            real qc[3], p0[3], pn[3], w[3];
            v3cpy(p0, q3); v3axpy(p0, -len3, L3);
            for (int it = 0; it < 48; it++) {
                closest_on_cyl(cc, a, rad, hl, p0, qc);
                closest_on_box(cb, (const real (*)[3])B, hb, qc, pn);
                v3sub(w, pn, p0);
                v3cpy(p0, pn);
                if (v3dot(w, w) < (real)1e-12) break;
            }
            v3sub(w, qc, p0);
            real len = v3norm(w);
            if (len > margin) return 0;
            if (len > (real)1e-6) v3set(out[0].n, w[0] / len, w[1] / len, w[2] / len);
            else { v3cpy(out[0].n, bn); len = 0; }          /* already touching: the winning axis, no depth */
            v3cpy(out[0].pa, qc);
            v3cpy(out[0].pb, p0);
            out[0].dist = len;
            return 1;
        }
    }
    int sel[4];
    int ns = reduce4((const real (*)[3])pts, sep, m, sel);
    for (int c = 0; c < ns; c++) {
        int i = sel[c];
        v3cpy(out[c].pa, pts[i]);
        v3cpy(out[c].pb, pts[i]); v3axpy(out[c].pb, -sep[i], n);
        v3cpy(out[c].n, n);
        out[c].dist = sep[i];
    }
    return ns;
}

int pmgo_cyl_box(const double cc[3], const double Rc[9], double rad, double hl, const double cb[3], const double Rb[9],
                 const double hb[3], double margin, double* out)
{
    real c1[3], c2[3], R1[9], R2[9], H[3];
    for (int i = 0; i < 3; i++) { c1[i] = (real)cc[i]; c2[i] = (real)cb[i]; H[i] = (real)hb[i]; }
    for (int i = 0; i < 9; i++) { R1[i] = (real)Rc[i]; R2[i] = (real)Rb[i]; }
    CPoint cp[4];
    int n = cyl_box(c1, R1, (real)rad, (real)hl, c2, R2, H, (real)margin, cp);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { out[10 * i + k] = cp[i].pa[k]; out[10 * i + 3 + k] = cp[i].pb[k]; out[10 * i + 6 + k] = cp[i].n[k]; }
        out[10 * i + 9] = cp[i].dist;
    }
    return n;
}

int pmgo_box_box(const double ca[3], const double Ra[9], const double ha[3], const double cb[3], const double Rb[9],
                 const double hb[3], double margin, double* out)
{
    real a[3], b[3], RA[9], RB[9], HA[3], HB[3];
    for (int i = 0; i < 3; i++) { a[i] = (real)ca[i]; b[i] = (real)cb[i]; HA[i] = (real)ha[i]; HB[i] = (real)hb[i]; }
    for (int i = 0; i < 9; i++) { RA[i] = (real)Ra[i]; RB[i] = (real)Rb[i]; }
    CPoint cp[4];
    int n = box_box(a, RA, HA, b, RB, HB, (real)margin, cp);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { out[10 * i + k] = cp[i].pa[k]; out[10 * i + 3 + k] = cp[i].pb[k]; out[10 * i + 6 + k] = cp[i].n[k]; }
        out[10 * i + 9] = cp[i].dist;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* world state                                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    real pos[3], quat[4], vel[3], omg[3];
} Block;

typedef struct {
    real q[NJ], qd[NJ];
    real motor_target[NJ];
    real motor_maximp[NJ];
    real ee_target[3];
    real joint_target[7];
    real rest_pose[7];
    real grip_target;
    int arm_enabled;
    int elapsed, reset_count;
    real goal[16];           /* static targets: stack = per block, rearrange = per target slot */
    int order[NBMAX];
    real base_target[3];
    int level;               /* active curriculum level / sub-goal index (nb-1 = the full goal) */
    int moved;               /* rearrange curriculum: bit i = block i has a target */
    double cur_prob[NBMAX + 1], cur_count[NBMAX + 1]; /* curriculum_prob, num_generated_goals_per_curriculum (chest: nb + 1 levels) */
    int cur_goal_step;       /* curriculum_goal_step */
    Block blk[NBMAX];
    real ws_m[NJ], ws_l[NJ]; /* last substep's motor / limit impulses (only read with the warm_start prior) */
    int wsc_n;               /* last substep's contacts: pair, point on A, normal impulse (contact_warm_start prior only) */
    int wsc_a[MAX_CONTACTS], wsc_b[MAX_CONTACTS];
    real wsc_p[MAX_CONTACTS][3], wsc_imp[MAX_CONTACTS];
    mt19937 rng;
} World;

struct pmgo_env {
    pmg_config cfg;
    pmg_dims dims;
    int nb;              /* dynamic objects */
    int grasping, has_obj, in_air, start_on_table;
    real tip_init[3], obj_lo[3], obj_hi[3], tgt_lo[3], tgt_hi[3];
    real ee_lo[3], ee_hi[3];
    real table_c[3], table_h[3], table_mu;
    real obj_z;
    int multi;                   /* multi-block observation layout (block_stack / block_rearrange) */
    int curriculum_update;       /* activate_curriculum_update() */
    double goals_per_curriculum; /* num_goals_to_generate // num_curriculum */
    int chest;                   /* -1 none, 0 front sliding door (chest_push), 1 up sliding lid (chest_pick_and_place) */
    int nwall;
    real wall_c[5][3], wall_h[5][3];   /* static chest walls, world frame (never rotated: chest.py:33) */
    real door_c0[3], door_h[3], door_axis[3], door_upper, door_mass, door_open;
    real handle_c[3], handle_R[9], handle_rad, handle_hl;   /* handle cylinder, relative to the door centre */
    real keypoint[3][3];
    int obj_cyl;                 /* the single free object is the slide puck (cylinder) */
    real obj_inertia[3], obj_half[3], obj_mu; /* principal inertia, half extents (cyl: r, r, h/2), friction */
    World* w;
    int nthreads;
    char err[256];
};
static char g_create_err[256];

/* body ids in contacts: 0..NBMAX-1 blocks, 100+L robot Bullet link L, -1 static */
#define BODY_STATIC (-1)
#define BODY_ROBOT(L) (100 + (L))
#define BODY_DOOR 50        /* the chest door: one prismatic DoF (door + handle), state in World.goal[0..2] */
#define DOOR_Q(w) ((w)->goal[0])
#define DOOR_QD(w) ((w)->goal[1])
#define DOOR_MOTOR(w) ((w)->goal[2])   /* 1 once _get_obs found the door open and latched the position motor */

typedef struct {
    int a, b;
    real pa[3], pb[3], n[3], dist, mu;
} Contact;

typedef struct {
    real Jr[NJ];          /* robot part of the row Jacobian */
    real dvr[NJ];         /* M^-1 Jr^T */
    int has_robot;
    int blk[2];           /* block ids (-1 none) */
    real Jl[2][3], Ja[2][3];
    real dl[2][3], da[2][3];
    real Jd, dd;          /* chest door: Jacobian on the joint velocity and its response Jd / m */
    real diag_inv, rhs, lo, hi, applied, mu;
    int fric_of;          /* index of the normal row for friction rows */
} Row;

static void block_R(const Block* b, real* R) { quat_to_R(b->quat, R); }

static void block_inv_inertia_apply(const pmgo_env* e, const Block* b, const real* t, real* out)
{
    real R[9], l[3];
    block_R(b, R);
    m3tv(l, R, t);
    for (int a = 0; a < 3; a++) l[a] /= e->obj_inertia[a];
    m3v(out, R, l);
}

/* fill a constraint row between body a (+) and body b (-) along direction n at points pa/pb */
static void row_setup(const pmgo_env* e, const World* w, const Kin* k, const AbaCache* ac, Row* r, int a, int b,
                      const real* pa, const real* pb, const real* n, real* rel_vel_out)
{
    memset(r, 0, sizeof(*r));
    r->blk[0] = r->blk[1] = -1;
    real denom = 0, rel = 0;
    int bodies[2] = {a, b};
    const real* pts[2] = {pa, pb};
    for (int s = 0; s < 2; s++) {
        real sg = s == 0 ? (real)1 : (real)-1;
        int id = bodies[s];
        if (id == BODY_STATIC) continue;
        if (id == BODY_DOOR) { /* a prismatic link on a fixed base: only the axis component of the direction acts */
            real jd = sg * v3dot(n, e->door_axis);
            r->Jd += jd;
            r->dd = r->Jd / e->door_mass;
            denom += jd * jd / e->door_mass;
            rel += jd * DOOR_QD(w);
            continue;
        }
        if (id >= 100) {
            real J[NJ], nn[3] = {sg * n[0], sg * n[1], sg * n[2]};
            point_jacobian(k, id - 100, pts[s], nn, J);
            for (int d = 0; d < NJ; d++) r->Jr[d] += J[d];
            r->has_robot = 1;
        } else {
            const Block* bl = &w->blk[id];
            real rp[3], rxn[3];
            v3sub(rp, pts[s], bl->pos);
            v3cross(rxn, rp, n);
            for (int c = 0; c < 3; c++) { r->Jl[s][c] = sg * n[c]; r->Ja[s][c] = sg * rxn[c]; }
            r->blk[s] = id;
            for (int c = 0; c < 3; c++) r->dl[s][c] = r->Jl[s][c] / (real)PMG_BLOCK_MASS;
            block_inv_inertia_apply(e, bl, r->Ja[s], r->da[s]);
            denom += v3dot(r->Jl[s], r->dl[s]) + v3dot(r->Ja[s], r->da[s]);
            rel += v3dot(r->Jl[s], bl->vel) + v3dot(r->Ja[s], bl->omg);
        }
    }
    if (r->has_robot) {
        aba_response(k, ac, r->Jr, r->dvr);
        for (int d = 0; d < NJ; d++) { denom += r->Jr[d] * r->dvr[d]; rel += r->Jr[d] * w->qd[d]; }
    }
    r->diag_inv = denom > (real)1.1920929e-07 ? 1 / denom : 0; /* [BULLET-PRIOR] d > SIMD_EPSILON */
    *rel_vel_out = rel;
}

/* ------------------------------------------------------------------ */
/* one 2 ms substep: collide, forward dynamics, PGS, integrate          */
/* [BULLET-PRIOR] btMultiBodyDynamicsWorld::internalSingleStepSimulation  */
/* ------------------------------------------------------------------ */
static void robot_box_pose(const Kin* k, int L, real* c, real* R)
{
    v3cpy(c, k->p[L]);
    memcpy(R, k->R[L], 9 * sizeof(real));
}

/* object b against a box B (table / finger): box-box, or cylinder-box for the slide puck.
 * obj_is_A: the object is body A of the pair (normal from B to A); otherwise the box is A. */
static int obj_vs_box(const pmgo_env* e, const World* w, const real (*Rb)[9], int b, int obj_is_A, const real* bc,
                      const real* bR, const real* bh, CPoint* cp)
{
    if (!e->obj_cyl)
        return obj_is_A ? box_box(w->blk[b].pos, Rb[b], e->obj_half, bc, bR, bh, CONTACT_MARGIN, cp)
                        : box_box(bc, bR, bh, w->blk[b].pos, Rb[b], e->obj_half, CONTACT_MARGIN, cp);
    int n = cyl_box(w->blk[b].pos, Rb[b], e->obj_half[0], e->obj_half[2], bc, bR, bh, CONTACT_MARGIN, cp);
    if (!obj_is_A)
        for (int c = 0; c < n; c++) { /* cyl_box reports the cylinder as A: swap roles */
            real t[3];
            v3cpy(t, cp[c].pa); v3cpy(cp[c].pa, cp[c].pb); v3cpy(cp[c].pb, t);
            v3set(cp[c].n, -cp[c].n[0], -cp[c].n[1], -cp[c].n[2]);
        }
    return n;
}

static int collide(const pmgo_env* e, const World* w, const Kin* k, Contact* out)
{
    int nc = 0;
    CPoint cp[4];
    real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    real fh[3] = {(real)FINGER_HALF[0], (real)FINGER_HALF[1], (real)FINGER_HALF[2]};
    real Rb[NBMAX][9];
    for (int b = 0; b < e->nb; b++) block_R(&w->blk[b], Rb[b]);
#define EMIT(A, B, MU)                                                                  \
    for (int c_ = 0; c_ < n_ && nc < MAX_CONTACTS; c_++) {                              \
        Contact* o = &out[nc++];                                                        \
        o->a = (A); o->b = (B); o->mu = (MU); o->dist = cp[c_].dist;                    \
        v3cpy(o->pa, cp[c_].pa); v3cpy(o->pb, cp[c_].pb); v3cpy(o->n, cp[c_].n);        \
    }
    /* object (A) x table (B); an object that has left the table: x the floor of the robot URDF (PLANE_SWITCH_Z above) */
    for (int b = 0; b < e->nb; b++) {
        const int on_floor = w->blk[b].pos[2] < (real)PLANE_SWITCH_Z;
        const real plane_c[3] = {0, 0, 0}, plane_h[3] = {(real)PLANE_HALF[0], (real)PLANE_HALF[1], (real)PLANE_HALF[2]};
        int n_ = obj_vs_box(e, w, (const real (*)[9])Rb, b, 1, on_floor ? plane_c : e->table_c, I3, on_floor ? plane_h : e->table_h, cp);
        EMIT(b, BODY_STATIC, e->obj_mu * (on_floor ? (real)PMG_PLANE_FRICTION : e->table_mu))
    }
    /* block x block */
    for (int b = 0; b < e->nb; b++)
        for (int c = b + 1; c < e->nb; c++) {
            real dd[3];
            v3sub(dd, w->blk[b].pos, w->blk[c].pos);
            if (v3dot(dd, dd) > (real)(0.06 * 0.06)) continue; /* bounding spheres: 2*sqrt(3)*0.015+margin < 0.06 */
            int n_ = box_box(w->blk[b].pos, Rb[b], e->obj_half, w->blk[c].pos, Rb[c], e->obj_half, CONTACT_MARGIN, cp);
            EMIT(b, c, e->obj_mu * e->obj_mu)
        }
    /* fingers (A) x objects (B), fingers x table */
    static const int FL[2] = {PMG_BL_FINGER1, PMG_BL_FINGER2};
    for (int f = 0; f < 2; f++) {
        real fc[3], fR[9];
        robot_box_pose(k, FL[f], fc, fR);
        for (int b = 0; b < e->nb; b++) {
            real dd[3];
            v3sub(dd, fc, w->blk[b].pos);
            if (v3dot(dd, dd) > (real)(0.08 * 0.08)) continue; /* finger 0.0431 + object <= 0.0317 + margin */
            int n_ = obj_vs_box(e, w, (const real (*)[9])Rb, b, 0, fc, fR, fh, cp);
            EMIT(BODY_ROBOT(FL[f]), b, (real)PMG_FINGER_FRICTION * e->obj_mu)
        }
        if (fc[2] - (real)0.0431 < e->table_c[2] + e->table_h[2] + CONTACT_MARGIN) {
            int n_ = box_box(fc, fR, fh, e->table_c, I3, e->table_h, CONTACT_MARGIN, cp);
            EMIT(BODY_ROBOT(FL[f]), BODY_STATIC, (real)PMG_FINGER_FRICTION * e->table_mu)
        }
    }
    /* gripper base cylinder (A, on link 7) x blocks (B); the puck never reaches it (tip z >= 0.175) */
    if (!e->obj_cyl)
        for (int b = 0; b < e->nb; b++) {
            const real* gc = k->p[PMG_BL_GBASE];
            real dd[3];
            v3sub(dd, gc, w->blk[b].pos);
            if (v3dot(dd, dd) > (real)(0.085 * 0.085)) continue; /* 0.0539 + 0.026 + margin */
            int n_ = cyl_box(gc, k->R[PMG_BL_GBASE], (real)PMG_GBASE_RADIUS, (real)PMG_GBASE_HALFLEN, w->blk[b].pos, Rb[b],
                             e->obj_half, CONTACT_MARGIN, cp);
            EMIT(BODY_ROBOT(PMG_BL_GBASE), b, GBASE_FRICTION * e->obj_mu)
        }
    /* chest (kuka_multi_step_base_env.py:97-110, chest.py): blocks, fingers and the gripper base against the static walls
     * and the door box; fingers against the door handle.  Not generated (build choice): door x walls / table
     * (the door cannot move along those normals), handle x blocks, handle x gripper base */
    if (e->chest >= 0) {
        real bc[5][3];
        const real* bh[5];
        int nbox = e->nwall + 1;
        for (int x = 0; x < e->nwall; x++) { v3cpy(bc[x], e->wall_c[x]); bh[x] = e->wall_h[x]; }
        v3cpy(bc[e->nwall], e->door_c0);
        v3axpy(bc[e->nwall], DOOR_Q(w), e->door_axis);
        bh[e->nwall] = e->door_h;
#define BOX_ID(x) ((x) == e->nwall ? BODY_DOOR : BODY_STATIC)
#define NEAR_BOX(P, RAD, x) (RFABS((P)[0] - bc[x][0]) <= bh[x][0] + (RAD) && RFABS((P)[1] - bc[x][1]) <= bh[x][1] + (RAD) && \
                             RFABS((P)[2] - bc[x][2]) <= bh[x][2] + (RAD))
        for (int b = 0; b < e->nb; b++)
            for (int x = 0; x < nbox; x++) {
                if (!NEAR_BOX(w->blk[b].pos, (real)0.026 + CONTACT_MARGIN, x)) continue;
                int n_ = box_box(w->blk[b].pos, Rb[b], e->obj_half, bc[x], I3, bh[x], CONTACT_MARGIN, cp);
                EMIT(b, BOX_ID(x), e->obj_mu * (real)PMG_CHEST_WALL_FRICTION)
            }
        for (int f = 0; f < 2; f++) {
            real fc[3], fR[9];
            robot_box_pose(k, FL[f], fc, fR);
            for (int x = 0; x < nbox; x++) {
                if (!NEAR_BOX(fc, (real)0.0431 + CONTACT_MARGIN, x)) continue;
                int n_ = box_box(fc, fR, fh, bc[x], I3, bh[x], CONTACT_MARGIN, cp);
                EMIT(BODY_ROBOT(FL[f]), BOX_ID(x), (real)PMG_FINGER_FRICTION * (real)PMG_CHEST_WALL_FRICTION)
            }
            real hc[3], dd[3];
            v3add(hc, bc[e->nwall], e->handle_c);
            v3sub(dd, fc, hc);
            real lim = (real)0.0431 + e->handle_hl + e->handle_rad + CONTACT_MARGIN;
            if (v3dot(dd, dd) <= lim * lim) {
                int n_ = cyl_box(hc, e->handle_R, e->handle_rad, e->handle_hl, fc, fR, fh, CONTACT_MARGIN, cp);
                for (int c2 = 0; c2 < n_; c2++) { /* cyl_box reports the cylinder as A: the finger is A here */
                    real t[3];
                    v3cpy(t, cp[c2].pa); v3cpy(cp[c2].pa, cp[c2].pb); v3cpy(cp[c2].pb, t);
                    v3set(cp[c2].n, -cp[c2].n[0], -cp[c2].n[1], -cp[c2].n[2]);
                }
                EMIT(BODY_ROBOT(FL[f]), BODY_DOOR, (real)PMG_FINGER_FRICTION * (real)PMG_CHEST_HANDLE_FRICTION)
            }
        }
        for (int x = 0; x < nbox; x++) {
            const real* gc = k->p[PMG_BL_GBASE];
            if (!NEAR_BOX(gc, (real)0.0539 + CONTACT_MARGIN, x)) continue;
            int n_ = cyl_box(gc, k->R[PMG_BL_GBASE], (real)PMG_GBASE_RADIUS, (real)PMG_GBASE_HALFLEN, bc[x], I3, bh[x],
                             CONTACT_MARGIN, cp);
            EMIT(BODY_ROBOT(PMG_BL_GBASE), BOX_ID(x), GBASE_FRICTION * (real)PMG_CHEST_WALL_FRICTION)
        }
#undef BOX_ID
#undef NEAR_BOX
    }
#undef EMIT
    return nc;
}

/* [BULLET-PRIOR] btPlaneSpace1 */
static void plane_space(const real* n, real* p, real* q)
{
    if (RFABS(n[2]) > (real)0.7071067811865475244008443621048490) {
        real a = n[1] * n[1] + n[2] * n[2];
        real k = 1 / RSQRT(a);
        v3set(p, 0, -n[2] * k, n[1] * k);
        v3set(q, a * k, -n[0] * p[2], n[0] * p[1]);
    } else {
        real a = n[0] * n[0] + n[1] * n[1];
        real k = 1 / RSQRT(a);
        v3set(p, -n[1] * k, n[0] * k, 0);
        v3set(q, -n[2] * p[1], n[2] * p[0], a * k);
    }
}

static real row_solve(Row* r, real* dqd, real (*dbl)[3], real (*dba)[3], real* ddoor)
{
    real dv = r->Jd * *ddoor;
    if (r->has_robot)
        for (int d = 0; d < NJ; d++) dv += r->Jr[d] * dqd[d];
    for (int s = 0; s < 2; s++)
        if (r->blk[s] >= 0) dv += v3dot(r->Jl[s], dbl[r->blk[s]]) + v3dot(r->Ja[s], dba[r->blk[s]]);
    real delta = r->rhs - dv * r->diag_inv;
    real sum = r->applied + delta;
    if (sum < r->lo) { delta = r->lo - r->applied; r->applied = r->lo; }
    else if (sum > r->hi) { delta = r->hi - r->applied; r->applied = r->hi; }
    else r->applied = sum;
    if (r->has_robot)
        for (int d = 0; d < NJ; d++) dqd[d] += r->dvr[d] * delta;
    *ddoor += r->dd * delta;
    for (int s = 0; s < 2; s++)
        if (r->blk[s] >= 0) { v3axpy(dbl[r->blk[s]], delta, r->dl[s]); v3axpy(dba[r->blk[s]], delta, r->da[s]); }
    return r->diag_inv != 0 ? delta / r->diag_inv : 0;
}

static void substep(const pmgo_env* e, World* w, const real* tau)
{
    const real dt = SUBSTEP_DT;
    Kin k;
    AbaCache ac;
    kinematics(w->q, &k);
    /* 1. collision detection at the current poses */
    Contact con[MAX_CONTACTS];
    int nc = collide(e, w, &k, con);
    /* 2. unconstrained velocity update */
    real qdd[NJ];
    aba(&k, w->qd, tau, qdd, &ac);
    for (int d = 0; d < NJ; d++) w->qd[d] += dt * qdd[d];
    for (int b = 0; b < e->nb; b++) {
        Block* bl = &w->blk[b];
        /* floating base with zero links: gravity, link damping, gyroscopic term */
        real kl = LINK_DAMPING * (1 + v3norm(bl->vel)), ka = LINK_DAMPING * (1 + v3norm(bl->omg));
        real R[9], wl[3], Iw[3], gy[3], tq[3], al[3];
        block_R(bl, R);
        m3tv(wl, R, bl->omg);
        for (int a = 0; a < 3; a++) Iw[a] = e->obj_inertia[a] * wl[a];
        v3cross(gy, wl, Iw);
        for (int a = 0; a < 3; a++) tq[a] = (-Iw[a] * ka - gy[a]) / e->obj_inertia[a];
        m3v(al, R, tq);
        for (int a = 0; a < 3; a++) {
            bl->vel[a] += dt * (-bl->vel[a] * kl + (a == 2 ? -GRAVITY : 0));
            bl->omg[a] += dt * al[a];
        }
    }
    /* 3. constraint rows */
    Row nonc[2 * NJ];
    int nn = 0;
    for (int oi = 0; oi < 18; oi++) {
        int ci = ROW_ORDER[oi];
        int d = ci % NJ;
        Row* r = &nonc[nn];
        if (ci >= NJ) {
            /* [BULLET-PRIOR] btMultiBodyJointMotor::createConstraintRows (erp 1, rhsClamp inf) */
            if (w->motor_maximp[d] <= 0) continue;
            memset(r, 0, sizeof(*r));
            r->blk[0] = r->blk[1] = -1;
            r->has_robot = 1;
            r->Jr[d] = 1;
            aba_response(&k, &ac, r->Jr, r->dvr);
            real den = r->dvr[d];
            r->diag_inv = den > (real)1.1920929e-07 ? 1 / den : 0;
            real kd = ARM_KD, kp = ARM_KP;
            real target_v = kp * (w->motor_target[d] - w->q[d]) / dt + w->qd[d] + kd * (0 - w->qd[d]);
            r->rhs = (target_v - w->qd[d]) * r->diag_inv;
            r->lo = -w->motor_maximp[d];
            r->hi = w->motor_maximp[d];
            nn++;
        } else {
            /* [BULLET-PRIOR] btMultiBodyJointLimitConstraint::createConstraintRows */
            for (int side = 0; side < 2; side++) {
                real pen = side == 0 ? w->q[d] - (real)JLO[d] : (real)JHI[d] - w->q[d];
                if (pen > 0) continue;
                r = &nonc[nn];
                memset(r, 0, sizeof(*r));
                r->blk[0] = r->blk[1] = -1;
                r->has_robot = 1;
                r->Jr[d] = side == 0 ? (real)1 : (real)-1;
                aba_response(&k, &ac, r->Jr, r->dvr);
                real den = r->Jr[d] * r->dvr[d];
                r->diag_inv = den > (real)1.1920929e-07 ? 1 / den : 0;
                real rel = r->Jr[d] * w->qd[d];
                r->rhs = (-pen * JOINT_ERP / dt - rel) * r->diag_inv;
                r->lo = 0;
                r->hi = LIMIT_MAX_IMPULSE;
                nn++;
            }
        }
    }
    /* chest door: joint limits ([BULLET-PRIOR] btMultiBodyJointLimitConstraint, as above) and, once latched by
     * _get_obs, the position motor of chest.py:59-68 (force 500, gains 0.03 / 1).  Solved after the robot's
     * non-contact rows, in this order, every iteration */
    Row drow[3];
    int nd = 0;
    if (e->chest >= 0) {
        real dm_inv = 1 / e->door_mass;
        if (DOOR_MOTOR(w) != 0) {
            Row* r = &drow[nd++];
            memset(r, 0, sizeof(*r));
            r->blk[0] = r->blk[1] = -1;
            r->Jd = 1; r->dd = dm_inv;
            r->diag_inv = e->door_mass;
            real target_v = ARM_KP * (e->door_open - DOOR_Q(w)) / dt + DOOR_QD(w) + ARM_KD * (0 - DOOR_QD(w));
            r->rhs = (target_v - DOOR_QD(w)) * r->diag_inv;
            r->lo = -(real)500.0 * PHYSICS_DT;
            r->hi = (real)500.0 * PHYSICS_DT;
        }
        for (int side = 0; side < 2; side++) {
            real pen = side == 0 ? DOOR_Q(w) : e->door_upper - DOOR_Q(w);
            if (pen > 0) continue;
            Row* r = &drow[nd++];
            memset(r, 0, sizeof(*r));
            r->blk[0] = r->blk[1] = -1;
            r->Jd = side == 0 ? (real)1 : (real)-1;
            r->dd = r->Jd * dm_inv;
            r->diag_inv = e->door_mass;
            r->rhs = (-pen * JOINT_ERP / dt - r->Jd * DOOR_QD(w)) * r->diag_inv;
            r->lo = 0;
            r->hi = LIMIT_MAX_IMPULSE;
        }
    }
    Row nrm[MAX_CONTACTS], fri[2 * MAX_CONTACTS];
    for (int c = 0; c < nc; c++) {
        /* [BULLET-PRIOR] btMultiBodyConstraintSolver::setupMultiBodyContactConstraint */
        Contact* cp = &con[c];
        real rel;
        Row* r = &nrm[c];
        row_setup(e, w, &k, &ac, r, cp->a, cp->b, cp->pa, cp->pb, cp->n, &rel);
        real dist = cp->dist + LINEAR_SLOP;
        real pos_err = 0, vel_err = -rel;
        if (dist > 0) vel_err -= dist / dt;
        else pos_err = -dist * CONTACT_ERP / dt;
        r->rhs = (pos_err + vel_err) * r->diag_inv;
        r->lo = 0;
        r->hi = (real)1e10;
        real t1[3], t2[3];
        plane_space(cp->n, t1, t2);
        real* tt[2] = {t1, t2};
        for (int f = 0; f < 2; f++) {
            if (f == 1 && G_PRIOR[PRIOR_FRICTION_DIRS] == 1) {   /* alternative prior: one friction row only */
                Row* fr = &fri[2 * c + f];
                memset(fr, 0, sizeof(*fr));
                fr->blk[0] = fr->blk[1] = -1;
                fr->fric_of = c;
                continue;
            }
            Row* fr = &fri[2 * c + f];
            row_setup(e, w, &k, &ac, fr, cp->a, cp->b, cp->pa, cp->pb, tt[f], &rel);
            fr->rhs = -rel * fr->diag_inv;
            fr->mu = cp->mu;
            fr->fric_of = c;
            fr->lo = fr->hi = 0;
        }
    }
    /* 4. projected Gauss-Seidel, [BULLET-PRIOR] btMultiBodyConstraintSolver::solveSingleIteration */
    real dqd[NJ], dbl[NBMAX][3], dba[NBMAX][3], ddoor = 0;
    memset(dqd, 0, sizeof(dqd));
    memset(dbl, 0, sizeof(dbl));
    memset(dba, 0, sizeof(dba));
    if (G_PRIOR[PRIOR_WARM_START] != 0)   /* alternative prior: the non-contact rows start from last substep's impulses */
        for (int j = 0; j < nn; j++) {
            Row* r = &nonc[j];
            int d = 0;
            while (d < NJ - 1 && r->Jr[d] == 0) d++;
            real a0 = (real)G_PRIOR[PRIOR_WARM_START] * (r->lo < 0 ? w->ws_m[d] : w->ws_l[d]);
            a0 = a0 < r->lo ? r->lo : (a0 > r->hi ? r->hi : a0);
            r->applied = a0;
            for (int k = 0; k < NJ; k++) dqd[k] += r->dvr[k] * a0;
        }
    if (G_PRIOR[PRIOR_CONTACT_WARM_START] != 0)   /* alternative prior: persistent contact points keep f x their impulse */
        for (int c = 0; c < nc; c++) {
            int best = -1;
            real bd = (real)(0.02 * 0.02);            /* btPersistentManifold contact breaking threshold */
            for (int j = 0; j < w->wsc_n; j++) {
                if (w->wsc_a[j] != con[c].a || w->wsc_b[j] != con[c].b) continue;
                real d[3];
                v3sub(d, w->wsc_p[j], con[c].pa);
                real dd = v3dot(d, d);
                if (dd < bd) { bd = dd; best = j; }
            }
            if (best < 0) continue;
            Row* r = &nrm[c];
            real a0 = (real)G_PRIOR[PRIOR_CONTACT_WARM_START] * w->wsc_imp[best];
            if (a0 <= 0) continue;
            w->wsc_a[best] = -12345;                   /* one cached point feeds one new point */
            r->applied = a0;
            if (r->has_robot)
                for (int d = 0; d < NJ; d++) dqd[d] += r->dvr[d] * a0;
            ddoor += r->dd * a0;
            for (int s2 = 0; s2 < 2; s2++)
                if (r->blk[s2] >= 0) { v3axpy(dbl[r->blk[s2]], a0, r->dl[s2]); v3axpy(dba[r->blk[s2]], a0, r->da[s2]); }
        }
    const int solver_iters = (int)G_PRIOR[PRIOR_SOLVER_ITERATIONS];
    for (int it = 0; it < solver_iters; it++) {
        real resid = 0;
        for (int j = 0; j < nn; j++) {
            int idx = (it & 1) ? j : nn - 1 - j;
            real dv = row_solve(&nonc[idx], dqd, dbl, dba, &ddoor);
            resid = dv * dv > resid ? dv * dv : resid;
        }
        for (int j = 0; j < nd; j++) {
            real dv = row_solve(&drow[j], dqd, dbl, dba, &ddoor);
            resid = dv * dv > resid ? dv * dv : resid;
        }
        for (int c = 0; c < nc; c++) {
            real dv = row_solve(&nrm[c], dqd, dbl, dba, &ddoor);
            resid = dv * dv > resid ? dv * dv : resid;
        }
        for (int c = 0; c < 2 * nc; c++) {
            Row* fr = &fri[c];
            real tot = nrm[fr->fric_of].applied;
            if (tot > 0) {
                fr->lo = -fr->mu * tot;
                fr->hi = fr->mu * tot;
                real dv = row_solve(fr, dqd, dbl, dba, &ddoor);
                resid = dv * dv > resid ? dv * dv : resid;
            }
        }
        if (resid <= RESIDUAL_THRESHOLD) break;
    }
    w->wsc_n = nc;
    for (int c = 0; c < nc; c++) {
        w->wsc_a[c] = con[c].a; w->wsc_b[c] = con[c].b; w->wsc_imp[c] = nrm[c].applied;
        v3cpy(w->wsc_p[c], con[c].pa);
    }
    for (int d = 0; d < NJ; d++) { w->ws_m[d] = 0; w->ws_l[d] = 0; }
    for (int j = 0; j < nn; j++) {
        int d = 0;
        while (d < NJ - 1 && nonc[j].Jr[d] == 0) d++;
        if (nonc[j].lo < 0) w->ws_m[d] = nonc[j].applied; else w->ws_l[d] = nonc[j].applied;
    }
    for (int d = 0; d < NJ; d++) w->qd[d] += dqd[d];
    /* 5. integrate positions */
    for (int d = 0; d < NJ; d++) w->q[d] += dt * w->qd[d];
    if (e->chest >= 0) { DOOR_QD(w) += ddoor; DOOR_Q(w) += dt * DOOR_QD(w); }
    for (int b = 0; b < e->nb; b++) {
        Block* bl = &w->blk[b];
        for (int a = 0; a < 3; a++) { bl->vel[a] += dbl[b][a]; bl->omg[a] += dba[b][a]; }
        for (int a = 0; a < 3; a++) bl->pos[a] += dt * bl->vel[a];
        /* quat <- exp(omega dt) * quat  ([BULLET-PRIOR] btMultiBody::stepPositionsMultiDof) */
        real ang = v3norm(bl->omg) * dt;
        real dq[4] = {0, 0, 0, 1};
        if (ang > (real)1e-12) {
            real s = RSIN(ang / 2) / (ang / dt);
            dq[0] = bl->omg[0] * s; dq[1] = bl->omg[1] * s; dq[2] = bl->omg[2] * s; dq[3] = RCOS(ang / 2);
        }
        real nq[4];
        quat_mul(nq, dq, bl->quat);
        real nn2 = RSQRT(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
        for (int a = 0; a < 4; a++) bl->quat[a] = nq[a] / nn2;
    }
    if (G_PRIOR[PRIOR_STATE_F32] != 0) {       /* the float32-state yardstick (see the table at the top) */
#define F32R(x) ((x) = (real)(float)(x))
        for (int d = 0; d < NJ; d++) { F32R(w->q[d]); F32R(w->qd[d]); }
        if (e->chest >= 0) { F32R(DOOR_Q(w)); F32R(DOOR_QD(w)); }
        for (int b = 0; b < e->nb; b++) {
            Block* bl = &w->blk[b];
            for (int a = 0; a < 3; a++) { F32R(bl->pos[a]); F32R(bl->vel[a]); F32R(bl->omg[a]); }
            for (int a = 0; a < 4; a++) F32R(bl->quat[a]);
        }
#undef F32R
    }
}

/* stepSimulation(): 20 substeps with forces latched at entry.
 * [BULLET-PRIOR] PhysicsServerCommandProcessor applies URDF joint damping once per
 * stepSimulation as a joint torque (-damping*qd) that persists over the substeps. */
static void step_simulation(const pmgo_env* e, World* w)
{
    real tau[NJ];
    for (int d = 0; d < NJ; d++) tau[d] = -(real)JDAMP[d] * w->qd[d];
    for (int s = 0; s < SUBSTEPS; s++) {
        if (G_PRIOR[PRIOR_DAMPING_PER_SUBSTEP] != 0)
            for (int d = 0; d < NJ; d++) tau[d] = -(real)JDAMP[d] * w->qd[d];
        substep(e, w, tau);
    }
}

/* ------------------------------------------------------------------ */
/* environment logic                                                    */
/* ------------------------------------------------------------------ */
static void env_constants(pmgo_env* e)
{
    const pmg_config* c = &e->cfg;
    int t = c->task;
    e->chest = t == PMG_TASK_CHEST_PUSH ? 0 : (t == PMG_TASK_CHEST_PICK_AND_PLACE ? 1 : -1);
    e->grasping = (t == PMG_TASK_PICK_AND_PLACE || t == PMG_TASK_BLOCK_STACK || t == PMG_TASK_CHEST_PICK_AND_PLACE);
    e->has_obj = (t != PMG_TASK_REACH);
    e->in_air = (t == PMG_TASK_REACH || t == PMG_TASK_PICK_AND_PLACE || t == PMG_TASK_BLOCK_STACK || t == PMG_TASK_CHEST_PICK_AND_PLACE);
    e->start_on_table = (t == PMG_TASK_PUSH || t == PMG_TASK_SLIDE || t == PMG_TASK_BLOCK_REARRANGE || t == PMG_TASK_CHEST_PUSH); /* kuka_multi_step_envs.py:169,250,399 */
    e->multi = (t == PMG_TASK_BLOCK_STACK || t == PMG_TASK_BLOCK_REARRANGE || e->chest >= 0);
    e->nb = t == PMG_TASK_REACH ? 0 : (e->multi ? c->num_block : 1);
    real obj_range = (t == PMG_TASK_SLIDE || e->chest >= 0) ? (real)0.1 : (real)0.15, /* kuka_multi_step_envs.py:251,400 */ tgt_range = t == PMG_TASK_SLIDE ? (real)0.2 : (real)0.15;
    /* kuka.py:35-51 */
    v3set(e->tip_init, (real)-0.52, 0, (real)0.25);
    if (e->start_on_table) e->tip_init[2] = (real)0.175 + (real)0.001;
    v3set(e->ee_hi, (real)-0.37, (real)0.20, (real)0.55);
    v3set(e->ee_lo, (real)-0.67, (real)-0.20, (real)0.175);
    for (int a = 0; a < 3; a++) {
        e->obj_lo[a] = e->tip_init[a] - obj_range; e->obj_hi[a] = e->tip_init[a] + obj_range;
        e->tgt_lo[a] = e->tip_init[a] - tgt_range; e->tgt_hi[a] = e->tip_init[a] + tgt_range;
    }
    e->obj_lo[0] += (real)0.03; e->obj_hi[0] -= (real)0.03;
    e->tgt_lo[0] += (real)0.03; e->tgt_hi[0] -= (real)0.03;
    e->tgt_lo[2] = e->ee_lo[2];
    if (e->chest >= 0) { /* kuka_multi_step_base_env.py:97-110 + the chest URDFs (include/pmg_model.h) */
        static const double CB[3] = PMG_CHEST_BASE, WC[2][4][3] = PMG_CHEST_WALL_C, WH[2][4][3] = PMG_CHEST_WALL_HALF;
        static const double DC[2][3] = PMG_CHEST_DOOR_C, DH[2][3] = PMG_CHEST_DOOR_HALF, DA[2][3] = PMG_CHEST_DOOR_AXIS;
        static const double DU[2] = PMG_CHEST_DOOR_UPPER, DM[2] = PMG_CHEST_DOOR_MASS, HC[2][3] = PMG_CHEST_HANDLE_C;
        static const double HR[2][3][3] = PMG_CHEST_HANDLE_R, HRAD[2] = PMG_CHEST_HANDLE_RADIUS, HHL[2] = PMG_CHEST_HANDLE_HALFLEN;
        static const double KP[2][3][3] = PMG_CHEST_KEYPOINTS;
        static const int NW[2] = PMG_CHEST_NWALL;
        int k = e->chest;
        e->obj_lo[0] += (real)0.05; e->obj_hi[0] += (real)0.05;
        e->obj_lo[1] -= (real)0.05; e->obj_hi[1] += (real)0.05;
        e->nwall = NW[k];
        for (int x = 0; x < e->nwall; x++)
            for (int a = 0; a < 3; a++) { e->wall_c[x][a] = (real)(CB[a] + WC[k][x][a]); e->wall_h[x][a] = (real)WH[k][x][a]; }
        for (int a = 0; a < 3; a++) {
            e->door_c0[a] = (real)(CB[a] + DC[k][a]); e->door_h[a] = (real)DH[k][a]; e->door_axis[a] = (real)DA[k][a];
            e->handle_c[a] = (real)HC[k][a];
            for (int b = 0; b < 3; b++) { e->handle_R[3 * a + b] = (real)HR[k][a][b]; e->keypoint[a][b] = (real)KP[k][a][b]; }
        }
        e->door_upper = (real)DU[k]; e->door_mass = (real)DM[k];
        e->handle_rad = (real)HRAD[k]; e->handle_hl = (real)HHL[k];
        e->door_open = e->grasping ? (real)0.1 : (real)0.12;   /* :106-109 */
    }
    v3set(e->table_c, (real)-0.52, 0, (real)0.08); /* kuka_single_step_base_env.py:49 */
    for (int a = 0; a < 3; a++) e->table_h[a] = (real)TABLE_HALF[a];
    e->table_mu = (real)PMG_TABLE_FRICTION;
    e->obj_z = (real)0.175;
    e->obj_cyl = 0;
    e->obj_mu = (real)PMG_BLOCK_FRICTION;
    for (int a = 0; a < 3; a++) { e->obj_inertia[a] = (real)BLOCK_INERTIA[a]; e->obj_half[a] = (real)BLOCK_HALF[a]; }
    if (t == PMG_TASK_SLIDE) { /* kuka_single_step_base_env.py:53-56,66-69 */
        e->tgt_lo[0] -= (real)0.4; e->tgt_hi[0] -= (real)0.4;
        e->table_c[0] = (real)-0.70;
        static const double LT[3] = PMG_LONG_TABLE_HALF;
        for (int a = 0; a < 3; a++) e->table_h[a] = (real)LT[a];
        e->table_mu = (real)PMG_LONG_TABLE_FRICTION;
        e->obj_z = (real)0.170;
        static const double PI[3] = PMG_PUCK_INERTIA, PH[3] = PMG_PUCK_HALF;   /* cylinder_bulk.urdf */
        e->obj_cyl = 1;
        e->obj_mu = (real)PMG_PUCK_FRICTION;
        for (int a = 0; a < 3; a++) { e->obj_inertia[a] = (real)PI[a]; e->obj_half[a] = (real)PH[a]; }
    }
}

static void tip_state(const World* w, const Kin* k, real* pos, real* vel, real* omg)
{
    v3cpy(pos, k->p[PMG_BL_TIP]);
    point_velocity(k, w->qd, PMG_BL_TIP, pos, vel, omg);
}

/* robot reset, kuka.py:120-165 */
static void robot_reset(const pmgo_env* e, World* w)
{
    real tq[4] = {(real)TOOL_QUAT[0], (real)TOOL_QUAT[1], (real)TOOL_QUAT[2], (real)TOOL_QUAT[3]};
    w->wsc_n = 0;
    for (int d = 0; d < 7; d++) { w->q[d] = w->rest_pose[d]; w->qd[d] = 0; w->motor_maximp[d] = 0; } /* :158 + robot_bases.py:230-238 */
    real qo[NJ];
    ik_solve(w->q, e->tip_init, tq, IK_MAX_ITER, IK_THRESHOLD, qo);                                    /* :159 */
    for (int d = 0; d < 7; d++) { w->rest_pose[d] = qo[d]; w->q[d] = qo[d]; w->qd[d] = 0; }          /* :160 */
    for (int d = 7; d < 9; d++) { w->q[d] = FINGER_LIMIT; w->qd[d] = 0; }                             /* :161 */
    w->grip_target = FINGER_LIMIT;                                                                     /* :162 */
    for (int d = 7; d < 9; d++) { w->motor_target[d] = FINGER_LIMIT; w->motor_maximp[d] = FINGER_FORCE * PHYSICS_DT; }
    w->arm_enabled = 0;
    Kin k;
    kinematics(w->q, &k);
    v3cpy(w->ee_target, k.p[PMG_BL_TIP]);                                                              /* :163 */
    for (int d = 0; d < 7; d++) w->joint_target[d] = w->q[d];                                          /* :165 */
}

static void set_block(Block* b, real x, real y, real z)
{
    v3set(b->pos, x, y, z);
    b->quat[0] = b->quat[1] = b->quat[2] = 0; b->quat[3] = 1;
    v3set(b->vel, 0, 0, 0);
    v3set(b->omg, 0, 0, 0);
}

static double norm2d(double x, double y) { return sqrt(x * x + y * y); } /* np.linalg.norm of a 2-vector */

/* single-step task reset: kuka_single_step_base_env.py:76-148 */
static void task_reset_single(const pmgo_env* e, World* w)
{
    double center[3] = {e->tip_init[0], e->tip_init[1], e->tip_init[2]};
    if (e->has_obj) {
        double oxy[2] = {e->tip_init[0], e->tip_init[1]};
        while (norm2d(oxy[0] - e->tip_init[0], oxy[1] - e->tip_init[1]) < 0.1) {      /* :108-111 */
            oxy[0] = mt_uniform(&w->rng, e->obj_lo[0], e->obj_hi[0]);
            oxy[1] = mt_uniform(&w->rng, e->obj_lo[1], e->obj_hi[1]);
        }
        set_block(&w->blk[0], (real)oxy[0], (real)oxy[1], e->obj_z);
        center[0] = oxy[0]; center[1] = oxy[1]; center[2] = e->obj_z;
    }
    double g[3];
    for (;;) {                                                                          /* :132-136 */
        for (int a = 0; a < 3; a++) g[a] = mt_uniform(&w->rng, e->tgt_lo[a], e->tgt_hi[a]);
        double dx = g[0] - center[0], dy = g[1] - center[1], dz = g[2] - center[2];
        if (sqrt(dx * dx + dy * dy + dz * dz) > 0.1) break;
    }
    if (!e->in_air) g[2] = e->obj_z;                                                    /* :138-139 */
    else if (e->grasping) {
        if (mt_uniform(&w->rng, 0, 1) >= 0.5) g[2] = e->obj_z;                          /* :140-143 */
    }
    for (int a = 0; a < 3; a++) w->goal[a] = (real)g[a];
}

/* block-stack reset: kuka_multi_step_base_env.py:221-250 + kuka_multi_step_envs.py:34-87 */
static void stack_goal_from_order(const pmgo_env* e, World* w)
{
    for (int s = 0; s < e->nb; s++) {
        int b = w->order[s];
        w->goal[3 * b] = w->base_target[0];
        w->goal[3 * b + 1] = w->base_target[1];
        w->goal[3 * b + 2] = (real)0.175 + (real)0.03 * (real)s;
    }
}
/* [NUMPY] RandomState.choice(n, p=p), size None: cdf = p.cumsum(); cdf /= cdf[-1];
 * idx = cdf.searchsorted(random_sample(), side='right')  (tests/golden/curriculum.json pins it) */
static int mt_choice_p(mt19937* rng, const double* p, int n)
{
    double cdf[NBMAX + 1], acc = 0;
    for (int i = 0; i < n; i++) { acc += p[i]; cdf[i] = acc; }
    for (int i = 0; i < n; i++) cdf[i] /= cdf[n - 1];
    double u = mt_double(rng);
    int idx = 0;
    while (idx < n && cdf[idx] <= u) idx++;
    return idx;
}

/* _update_curriculum_prob: kuka_multi_step_base_env.py:350-379 */
/* num_curriculum: num_block (block_stack / block_rearrange, kuka_multi_step_envs.py:19,166), num_block + 1 (chest
 * tasks, :253,402: level = how many blocks go into the chest, 0 = just open the door) */
static int num_curriculum(const pmgo_env* e) { return e->chest >= 0 ? e->nb + 1 : e->nb; }
static void update_curriculum_prob(const pmgo_env* e, World* w)
{
    int n = num_curriculum(e);
    int fin[NBMAX + 1], half[NBMAX + 1];
    for (int i = 0; i < n; i++) {
        fin[i] = w->cur_count[i] >= e->goals_per_curriculum;
        half[i] = w->cur_count[i] >= e->goals_per_curriculum / 2;
        if (fin[i]) w->cur_prob[i] = 0.0;
    }
    if (half[0] && !fin[0]) { w->cur_prob[0] = 0.5; w->cur_prob[1] = 0.5; }
    for (int i = 1; i < n - 1; i++)
        if (fin[i - 1] && !fin[i]) {
            if (half[i]) { w->cur_prob[i] = 0.5; w->cur_prob[i + 1] = 0.5; }
            else w->cur_prob[i] = 1.0;
        }
    if (fin[n - 2]) w->cur_prob[n - 1] = 1.0;
}

/* curriculum level draw shared by both tasks: kuka_multi_step_envs.py:125-134, 199-213 */
static void curriculum_new_target(const pmgo_env* e, World* w)
{
    int level = mt_choice_p(&w->rng, w->cur_prob, num_curriculum(e));
    w->level = level;
    w->cur_goal_step = level * 25 + 50;
    if (e->chest >= 0) {
        /* kuka_multi_step_envs.py:351-357, 484-490: choice(arange(nb), size=level, replace=False) == permutation(nb)[:level]
         * (the permutation is drawn even for size 0) */
        int perm[NBMAX];
        for (int i = 0; i < e->nb; i++) perm[i] = i;
        for (int i = e->nb - 1; i >= 1; i--) {
            uint32_t j = mt_interval(&w->rng, (uint32_t)i);
            int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        w->moved = 0;
        for (int i = 0; i < level; i++) w->moved |= 1 << perm[i];
    }
    if (e->cfg.task == PMG_TASK_BLOCK_REARRANGE) {
        /* np_random.choice(arange(nb), size=level+1, replace=False) == permutation(nb)[:level+1] (sorted after) */
        int perm[NBMAX];
        for (int i = 0; i < e->nb; i++) perm[i] = i;
        for (int i = e->nb - 1; i >= 1; i--) {
            uint32_t j = mt_interval(&w->rng, (uint32_t)i);
            int t = perm[i]; perm[i] = perm[j]; perm[j] = t;
        }
        w->moved = 0;
        for (int i = 0; i <= level; i++) w->moved |= 1 << perm[i];
    }
    if (e->curriculum_update) {
        w->cur_count[level] += 1;
        update_curriculum_prob(e, w);
    }
}

/* num_steps of the chest tasks: kuka_multi_step_envs.py:238-242, 388-392 */
static int chest_num_steps(const pmgo_env* e)
{
    if (!e->cfg.grip_informed_goal) return e->nb + 1;
    return e->nb * (e->grasping ? 3 : 2) + 1;
}

/* multi-block reset: kuka_multi_step_base_env.py:221-250 + kuka_multi_step_envs.py:34-87 (stack),
 * :174-197 (rearrange) */
static void task_reset_multi(const pmgo_env* e, World* w)
{
    double bp[NBMAX][2];
    for (int b = 0; b < e->nb; b++) {
        for (;;) {
            double x = mt_uniform(&w->rng, e->obj_lo[0], e->obj_hi[0]);
            double y = mt_uniform(&w->rng, e->obj_lo[1], e->obj_hi[1]);
            int ok = 1;
            for (int c = 0; c < b; c++)
                if (!(norm2d(x - bp[c][0], y - bp[c][1]) > 0.06)) ok = 0;
            if (!(norm2d(x - e->tip_init[0], y - e->tip_init[1]) > 0.06)) ok = 0;
            if (ok) { bp[b][0] = x; bp[b][1] = y; break; }
        }
    }
    for (int b = 0; b < e->nb; b++) set_block(&w->blk[b], (real)bp[b][0], (real)bp[b][1], (real)0.175);
    for (int b = 0; b < NBMAX; b++) w->order[b] = b;
    /* sub_goal_ind = -1 after reset (kuka_multi_step_base_env.py:248-249): the last sub-goal */
    w->level = (e->cfg.grip_informed_goal && e->cfg.task_decomposition) ? 2 * e->nb - 1 : e->nb - 1;
    w->moved = (1 << e->nb) - 1;
    if (e->chest >= 0) {
        /* chest_robot.robot_specific_reset (:242-243, chest.py:40-45): door closed, at rest, motor off.  No random
         * target: the goal is the chest (kuka_multi_step_envs.py:256-283, 405-431).  sub_goal_ind = -1 = the last step */
        for (int g = 0; g < 16; g++) w->goal[g] = 0;
        w->level = chest_num_steps(e) - 1;
        if (e->cfg.use_curriculum) curriculum_new_target(e, w);
        return;
    }
    if (e->cfg.task == PMG_TASK_BLOCK_STACK) {
        if (e->cfg.random_order)
            for (int i = e->nb - 1; i >= 1; i--) {
                uint32_t j = mt_interval(&w->rng, (uint32_t)i);
                int t = w->order[i]; w->order[i] = w->order[j]; w->order[j] = t;
            }
        for (;;) {
            double x = mt_uniform(&w->rng, e->tgt_lo[0], e->tgt_hi[0]);
            double y = mt_uniform(&w->rng, e->tgt_lo[1], e->tgt_hi[1]);
            int ok = 1;
            for (int c = 0; c < e->nb; c++)
                if (!(norm2d(x - bp[c][0], y - bp[c][1]) > 0.08)) ok = 0;
            if (ok) { v3set(w->base_target, (real)x, (real)y, (real)0.175); break; }
        }
        stack_goal_from_order(e, w);
    } else {
        double tp[NBMAX][2];
        for (int t = 0; t < e->nb; t++)
            for (;;) {
                double x = mt_uniform(&w->rng, e->tgt_lo[0], e->tgt_hi[0]);
                double y = mt_uniform(&w->rng, e->tgt_lo[1], e->tgt_hi[1]);
                int ok = 1;
                for (int c = 0; c < t; c++)
                    if (!(norm2d(x - tp[c][0], y - tp[c][1]) > 0.06)) ok = 0;
                for (int c = 0; c < e->nb; c++)
                    if (!(norm2d(x - bp[c][0], y - bp[c][1]) > 0.06)) ok = 0;
                if (ok) { tp[t][0] = x; tp[t][1] = y; break; }
            }
        for (int t = 0; t < e->nb; t++) { w->goal[3 * t] = (real)tp[t][0]; w->goal[3 * t + 1] = (real)tp[t][1]; w->goal[3 * t + 2] = (real)0.175; }
    }
    if (e->cfg.use_curriculum) curriculum_new_target(e, w);
}

/* the desired goal as _get_obs re-derives it from the current block poses every observation
 * (kuka_multi_step_base_env.py:309-312 -> _generate_goal(new_target=False) / set_sub_goal) */
static void chest_goal(const pmgo_env* e, const World* w, double* dg);
static void effective_goal(const pmgo_env* e, const World* w, double* dg)
{
    int G = e->dims.goal_dim;
    if (!e->multi) { for (int g = 0; g < G; g++) dg[g] = w->goal[g]; return; }
    if (e->chest >= 0) { chest_goal(e, w, dg); return; }
    if (e->cfg.task == PMG_TASK_BLOCK_STACK) {
        /* plain / curriculum: the first level+1 blocks of the order sit at their targets.  grip-informed sub-goals
         * come in (pick, place) pairs per block j (kuka_multi_step_envs.py:91-111): pick keeps blocks i < j, place
         * blocks i <= j at their targets */
        int pairs = e->cfg.grip_informed_goal && e->cfg.task_decomposition;
        int j = pairs ? w->level / 2 : w->level, pick = pairs && (w->level % 2 == 0);
        for (int s_ = 0; s_ < e->nb; s_++) {
            int b = w->order[s_];
            int at_target = pick ? s_ < j : s_ <= j;
            for (int a = 0; a < 3; a++) dg[3 * b + a] = at_target ? w->goal[3 * b + a] : w->blk[b].pos[a];
        }
        if (e->cfg.grip_informed_goal) { /* gripper tip target + finger width 0.03 (:75-77, :98-99, :108-109, :143-145) */
            int bj = w->order[j];
            for (int a = 0; a < 3; a++) dg[3 * e->nb + a] = pick ? w->blk[bj].pos[a] : w->goal[3 * bj + a];
            dg[3 * e->nb + 3] = 0.03;
        }
    } else {
        int k = 0;
        for (int b = 0; b < e->nb; b++) {
            int mv = (w->moved >> b) & 1;
            for (int a = 0; a < 3; a++) dg[3 * b + a] = mv ? w->goal[3 * k + a] : w->blk[b].pos[a];
            k += mv;
        }
    }
}

/* chest tasks: _generate_goal / _generate_subgoals of kuka_multi_step_envs.py:256-342 (pick and place), 405-475 (push).
 * Element 0 is the door joint's open state; blocks in the chest sit at its floor centre, the others where they are now.
 * Sub-goal 0 (open the door) of the reference also appends the tip position (and finger width) WITHOUT
 * grip_informed_goal, which makes its length differ from achieved_goal's (the reference then fails its own assert,
 * kuka_multi_step_base_env.py:314): here the vector is cut to goal_dim. */
static void chest_goal(const pmgo_env* e, const World* w, double* dg)
{
    const double centre[3] = {-0.7 + 0.05, 0.0, 0.175}, top[3] = {-0.7 + 0.05, 0.0, 0.3};
    int nb = e->nb, pnp = e->grasping, grip = e->cfg.grip_informed_goal;
    int in_chest = (1 << nb) - 1, lifted = -1;        /* bit b: block b's goal is the chest; lifted: block held above it */
    int tip_goal = 0;                                 /* gripper goal = the current tip pose */
    double gx[3], gw = 0.06;
    for (int a = 0; a < 3; a++) gx[a] = pnp ? top[a] : centre[a] + (a == 0 ? 0.03 : 0.0);
    if (e->cfg.task_decomposition) {
        int ind = w->level;
        if (ind == 0) { in_chest = 0; tip_goal = 1; }
        else if (!grip) in_chest = (1 << ind) - 1;     /* blocks i <= ind-1 */
        else if (!pnp) {
            int j = (ind - 1) / 2, ph = (ind - 1) % 2;
            in_chest = (1 << (j + ph)) - 1;
            if (ph == 0) for (int a = 0; a < 3; a++) gx[a] = w->blk[j].pos[a] + (a == 0 ? (real)0.03 : 0);
        } else {
            int j = (ind - 1) / 3, ph = (ind - 1) % 3;
            in_chest = (1 << (j + (ph == 2))) - 1;
            if (ph == 0) for (int a = 0; a < 3; a++) gx[a] = w->blk[j].pos[a];
            if (ph == 1) lifted = j;
            gw = ph == 2 ? 0.06 : 0.03;
        }
    } else if (e->cfg.use_curriculum) { /* :359-381, 492-515: the chosen blocks go in; level 0 leaves the gripper where it is */
        in_chest = w->moved;
        tip_goal = w->level == 0;
    }
    dg[0] = e->door_open;
    for (int b = 0; b < nb; b++)
        for (int a = 0; a < 3; a++)
            dg[1 + 3 * b + a] = b == lifted ? top[a] : (((in_chest >> b) & 1) ? centre[a] : (double)w->blk[b].pos[a]);
    if (grip) {
        if (tip_goal) {
            Kin k;
            kinematics(w->q, &k);
            real d[3];
            v3sub(d, k.p[PMG_BL_TAB1], k.p[PMG_BL_TAB2]);
            for (int a = 0; a < 3; a++) gx[a] = k.p[PMG_BL_TIP][a];
            gw = v3norm(d);
        }
        for (int a = 0; a < 3; a++) dg[1 + 3 * nb + a] = gx[a];
        if (pnp) dg[1 + 3 * nb + 3] = gw;
    }
}

static void env_reset_one(const pmgo_env* e, World* w)
{
    /* the reference's constructor resets the robot once on its own (base_env.py:42) before its first env.reset()
     * (base_env.py:84): the very first reset of a world therefore runs the reset IK twice, each from the previous
     * solution (kuka.py:159-160) -- pinned by tests/golden/ref_*.json */
    if (w->reset_count == 0) robot_reset(e, w);
    robot_reset(e, w);
    if (e->multi) task_reset_multi(e, w);
    else task_reset_single(e, w);
    w->elapsed = 0;
    w->reset_count++;
}

/* observation assembly.  single: kuka_single_step_base_env.py:193-221; stack:
 * kuka_multi_step_base_env.py:255-336; robot state: kuka.py:227-256 */
static void env_obs(const pmgo_env* e, const World* w, float* obs, float* pol, float* ag, float* dg)
{
    Kin k;
    kinematics(w->q, &k);
    real tip[3], tv[3], tw[3];
    tip_state(w, &k, tip, tv, tw);
    real closeness = 0, fvel = 0;
    if (e->grasping) {
        real d[3];
        v3sub(d, k.p[PMG_BL_TAB1], k.p[PMG_BL_TAB2]);
        closeness = v3norm(d);
        real vb[3], vt[3], tmp[3];
        point_velocity(&k, w->qd, PMG_BL_GBASE, k.p[PMG_BL_GBASE], vb, tmp);
        point_velocity(&k, w->qd, PMG_BL_TAB1, k.p[PMG_BL_TAB1], vt, tmp);
        fvel = vb[1] - vt[1];
    }
    int jo = e->cfg.joint_control ? 7 : 0;
    int G = e->dims.goal_dim;
    double o[160], p[96];
    int no = 0, np = 0;
    if (jo) for (int d = 0; d < 7; d++) { o[no++] = w->q[d]; p[np++] = w->q[d]; }
    if (e->multi) {
        for (int a = 0; a < 3; a++) { o[no++] = tip[a]; p[np++] = tip[a]; }
        o[no++] = closeness; p[np++] = closeness;
        for (int a = 0; a < 3; a++) o[no++] = tv[a];
        o[no++] = fvel;
        for (int b = 0; b < e->nb; b++) {
            const Block* bl = &w->blk[b];
            for (int a = 0; a < 3; a++) o[no++] = bl->pos[a];
            for (int a = 0; a < 3; a++) { o[no++] = tip[a] - bl->pos[a]; p[np++] = tip[a] - bl->pos[a]; }
            for (int a = 0; a < 4; a++) o[no++] = bl->quat[a];
            for (int a = 0; a < 3; a++) o[no++] = tv[a] - bl->vel[a];
            for (int a = 0; a < 3; a++) o[no++] = tw[a] - bl->omg[a];
            if (ag) for (int a = 0; a < 3; a++) ag[(e->chest >= 0) + 3 * b + a] = (float)bl->pos[a];
        }
        if (e->chest >= 0) { /* :289-298 + chest.py:47-57: door joint state, then xyz + velocity of the three key points */
            o[no++] = DOOR_Q(w); o[no++] = DOOR_QD(w);
            p[np++] = DOOR_Q(w);
            for (int kp = 0; kp < 3; kp++) {
                for (int a = 0; a < 3; a++) { double x = e->door_c0[a] + e->door_axis[a] * DOOR_Q(w) + e->keypoint[kp][a]; o[no++] = x; p[np++] = x; }
                for (int a = 0; a < 3; a++) { double v = e->door_axis[a] * DOOR_QD(w); o[no++] = v; p[np++] = v; }
            }
            if (ag) ag[0] = (float)DOOR_Q(w);
        }
        if (ag && e->cfg.grip_informed_goal) { /* :300-304 */
            int g0 = (e->chest >= 0) + 3 * e->nb;
            for (int a = 0; a < 3; a++) ag[g0 + a] = (float)tip[a];
            if (e->grasping) ag[g0 + 3] = (float)closeness;
        }
        for (int i = 0; i < no; i++) o[i] = o[i] < -5 ? -5 : (o[i] > 5 ? 5 : o[i]);  /* :306-307 */
        for (int i = 0; i < np; i++) p[i] = p[i] < -5 ? -5 : (p[i] > 5 ? 5 : p[i]);
    } else if (e->has_obj) {
        const Block* bl = &w->blk[0];
        for (int a = 0; a < 3; a++) o[no++] = tip[a];
        for (int a = 0; a < 3; a++) o[no++] = bl->pos[a];
        o[no++] = closeness;
        for (int a = 0; a < 3; a++) o[no++] = tip[a] - bl->pos[a];
        for (int a = 0; a < 3; a++) o[no++] = tv[a];
        o[no++] = fvel;
        for (int a = 0; a < 3; a++) o[no++] = tv[a] - bl->vel[a];
        for (int a = 0; a < 3; a++) o[no++] = tw[a] - bl->omg[a];
        for (int a = 0; a < 3; a++) p[np++] = tip[a];
        p[np++] = closeness;
        for (int a = 0; a < 3; a++) p[np++] = tip[a] - bl->pos[a];
        if (ag) for (int a = 0; a < 3; a++) ag[a] = (float)bl->pos[a];
    } else {
        for (int a = 0; a < 3; a++) { o[no++] = tip[a]; p[np++] = tip[a]; }
        if (ag) for (int a = 0; a < 3; a++) ag[a] = (float)tip[a];
    }
    if (obs) for (int i = 0; i < no; i++) obs[i] = (float)o[i];
    if (pol) for (int i = 0; i < np; i++) pol[i] = (float)p[i];
    if (dg) {
        double d64[20];
        effective_goal(e, w, d64);
        for (int i = 0; i < G; i++) dg[i] = (float)d64[i];
    }
}

/* reward: kuka_single_step_base_env.py:237-244 */
static void reward_f64(const pmgo_env* e, const double* ag, const double* dg, int G, float* r, uint8_t* ok)
{
    double s = 0;
    for (int i = 0; i < G; i++) s += (ag[i] - dg[i]) * (ag[i] - dg[i]);
    double d = sqrt(s);
    int not_achieved = d > (double)e->cfg.distance_threshold;
    if (e->cfg.binary_reward) *r = -(float)not_achieved;
    else *r = (float)(-d);
    *ok = (uint8_t)!not_achieved;
}

/* apply_action: kuka.py:167-225 */
static void env_step_one(const pmgo_env* e, World* w, const float* a)
{
    int A = e->dims.action_dim;
    real tq[4] = {(real)TOOL_QUAT[0], (real)TOOL_QUAT[1], (real)TOOL_QUAT[2], (real)TOOL_QUAT[3]};
    if (e->grasping) {
        w->grip_target = (real)(((double)a[A - 1] + 1.0) * (0.035 / 2));               /* :171 */
        for (int d = 7; d < 9; d++) { w->motor_target[d] = w->grip_target; w->motor_maximp[d] = FINGER_FORCE * PHYSICS_DT; }
    }
    real poses[NJ];
    if (e->cfg.joint_control) {
        for (int d = 0; d < 7; d++) { w->joint_target[d] = (real)(float)(a[d] * 0.05f) + w->joint_target[d]; poses[d] = w->joint_target[d]; } /* :205 */
    } else {
        for (int c = 0; c < 3; c++) {
            real t = w->ee_target[c] + (real)(float)(a[c] * 0.01f);                     /* :209 (float32 product) */
            w->ee_target[c] = t < e->ee_lo[c] ? e->ee_lo[c] : (t > e->ee_hi[c] ? e->ee_hi[c] : t); /* :210-212 */
        }
        ik_solve(w->q, w->ee_target, tq, IK_MAX_ITER, IK_THRESHOLD, poses);            /* :214 */
    }
    for (int d = 0; d < 7; d++) { w->motor_target[d] = poses[d]; w->motor_maximp[d] = ARM_FORCE * PHYSICS_DT; } /* :222, :282-290 */
    w->arm_enabled = 1;
    for (int s = 0; s < SIM_STEPS; s++) step_simulation(e, w);                         /* :223-225 */
    /* _get_obs keeps a door it finds open open: from now on the position motor holds it (:296-298) */
    if (e->chest >= 0 && RFABS(e->door_open - DOOR_Q(w)) <= (real)0.01) DOOR_MOTOR(w) = 1;
    w->elapsed++;
}

/* ------------------------------------------------------------------ */
/* C ABI                                                                */
/* ------------------------------------------------------------------ */
static int fill_dims(const pmg_config* c, pmg_dims* d)
{
    memset(d, 0, sizeof(*d));
    int jo = c->joint_control ? 7 : 0;
    d->num_envs = c->num_envs;
    switch (c->task) {
    case PMG_TASK_REACH:
        d->action_dim = jo ? 7 : 3; d->observation_dim = 3 + jo; d->policy_state_dim = 3 + jo; d->goal_dim = 3; break;
    case PMG_TASK_PUSH:
    case PMG_TASK_SLIDE:
        d->action_dim = jo ? 7 : 3; d->observation_dim = 20 + jo; d->policy_state_dim = 7 + jo; d->goal_dim = 3; break;
    case PMG_TASK_PICK_AND_PLACE:
        d->action_dim = jo ? 8 : 4; d->observation_dim = 20 + jo; d->policy_state_dim = 7 + jo; d->goal_dim = 3; break;
    case PMG_TASK_BLOCK_STACK:
    case PMG_TASK_BLOCK_REARRANGE: {
        if (c->num_block < 1 || c->num_block > NBMAX) return -1;
        if (c->use_curriculum && (c->num_block < 2 || c->task_decomposition)) return -1; /* kuka_multi_step_base_env.py:123,131 */
        if (c->task_decomposition && c->task != PMG_TASK_BLOCK_STACK) return -1;          /* kuka_multi_step_envs.py:159 */
        int gr = c->task == PMG_TASK_BLOCK_STACK;
        d->action_dim = (jo ? 7 : 3) + gr; d->observation_dim = 8 + 16 * c->num_block + jo;
        d->policy_state_dim = 4 + 3 * c->num_block + jo; d->goal_dim = 3 * c->num_block; 
        if (c->grip_informed_goal) {
            if (!gr) return -1;                       /* kuka_multi_step_envs.py:158: not for block_rearrange */
            d->goal_dim += 4;                         /* tip xyz + finger width, kuka_multi_step_base_env.py:300-304 */
        }
        break;
    }
    case PMG_TASK_CHEST_PUSH:
    case PMG_TASK_CHEST_PICK_AND_PLACE: {
        /* kuka_multi_step_base_env.py:283-304: the multi-block layout + door joint pos / vel + 3 key points x (xyz, vel);
         * goals lead with the door state */
        if (c->num_block < 1 || c->num_block > NBMAX || (c->use_curriculum && c->task_decomposition)) return -1;
        int gr = c->task == PMG_TASK_CHEST_PICK_AND_PLACE;
        d->action_dim = (jo ? 7 : 3) + gr; d->observation_dim = 8 + 16 * c->num_block + jo + 20;
        d->policy_state_dim = 4 + 3 * c->num_block + jo + 19; d->goal_dim = 1 + 3 * c->num_block;
        if (c->grip_informed_goal) d->goal_dim += gr ? 4 : 3;   /* :300-304 */
        break;
    }
    default: return -1;
    }
    int chest = c->task == PMG_TASK_CHEST_PUSH || c->task == PMG_TASK_CHEST_PICK_AND_PLACE;
    if (c->task != PMG_TASK_BLOCK_STACK && c->task != PMG_TASK_BLOCK_REARRANGE && !chest && (c->use_curriculum || c->task_decomposition || c->grip_informed_goal)) return -1;
    int multi = c->task == PMG_TASK_BLOCK_STACK || c->task == PMG_TASK_BLOCK_REARRANGE || chest;
    int nb = c->task == PMG_TASK_REACH ? 0 : (multi ? c->num_block : 1);
    d->state_dim = 64 + 13 * nb + (c->use_curriculum ? 16 : 0);
    d->packed_dim = d->observation_dim + d->policy_state_dim + 2 * d->goal_dim + 3;
    return 0;
}

int pmgo_create(const pmg_config* cfg, pmgo_env** out)
{
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(pmg_config) || cfg->num_envs < 1) {
        snprintf(g_create_err, sizeof(g_create_err), "pmgo_create: bad config");
        return PMG_E_INVALID;
    }
    pmgo_env* e = (pmgo_env*)calloc(1, sizeof(pmgo_env));
    e->cfg = *cfg;
    if (fill_dims(cfg, &e->dims) != 0) {
        snprintf(g_create_err, sizeof(g_create_err), "pmgo_create: unsupported task %d / num_block %d", cfg->task, cfg->num_block);
        free(e);
        return PMG_E_INVALID;
    }
    env_constants(e);
    e->w = (World*)calloc((size_t)cfg->num_envs, sizeof(World));
    e->nthreads = 1;
    for (int i = 0; i < cfg->num_envs; i++) {
        World* w = &e->w[i];
        for (int d = 0; d < 7; d++) w->rest_pose[d] = (real)REST_POSE0[d];
        for (int b = 0; b < NBMAX; b++) set_block(&w->blk[b], 0, 0, -3);
        w->cur_prob[0] = 1.0;              /* kuka_multi_step_base_env.py:133 */
        w->cur_goal_step = 50;
        w->level = 0;
    }
    {
        double total = cfg->num_goals_to_generate > 0 ? (double)cfg->num_goals_to_generate : 1e6;
        e->goals_per_curriculum = e->nb > 0 ? floor(total / num_curriculum(e)) : total;   /* :139 */
    }
    *out = e;
    pmgo_seed(e, cfg->seed_base, cfg->seed_stride);
    return PMG_OK;
}
void pmgo_destroy(pmgo_env* e)
{
    if (!e) return;
    free(e->w);
    free(e);
}
int pmgo_get_dims(const pmgo_env* e, pmg_dims* out) { *out = e->dims; return PMG_OK; }
const char* pmgo_last_error(const pmgo_env* e) { return e ? e->err : g_create_err; }
int pmgo_set_threads(pmgo_env* e, int n) { e->nthreads = n < 1 ? 1 : n; return PMG_OK; }

int pmgo_seed(pmgo_env* e, uint64_t base, uint64_t stride)
{
    e->cfg.seed_base = base;
    e->cfg.seed_stride = stride;
    for (int i = 0; i < e->cfg.num_envs; i++)
        gym_seed(&e->w[i].rng, base + stride * (uint64_t)(i + e->cfg.env_index_offset));
    return PMG_OK;
}

static void outputs(pmgo_env* e, int i, float* obs, float* pol, float* ag, float* dg)
{
    const pmg_dims* d = &e->dims;
    env_obs(e, &e->w[i], obs ? obs + (size_t)i * d->observation_dim : NULL, pol ? pol + (size_t)i * d->policy_state_dim : NULL,
            ag ? ag + (size_t)i * d->goal_dim : NULL, dg ? dg + (size_t)i * d->goal_dim : NULL);
}

int pmgo_reset(pmgo_env* e, const uint8_t* mask, float* obs, float* pol, float* ag, float* dg)
{
    int N = e->cfg.num_envs;
#pragma omp parallel for num_threads(e->nthreads) schedule(static)
    for (int i = 0; i < N; i++) {
        if (!mask || mask[i]) env_reset_one(e, &e->w[i]);
        outputs(e, i, obs, pol, ag, dg);
    }
    return PMG_OK;
}

int pmgo_step(pmgo_env* e, const float* actions, float* obs, float* pol, float* ag, float* dg, float* reward,
              uint8_t* goal_achieved, uint8_t* done)
{
    int N = e->cfg.num_envs;
    const pmg_dims* dm = &e->dims;
    for (int i = 0; i < N; i++)
        if (e->w[i].reset_count == 0) { snprintf(e->err, sizeof(e->err), "pmgo_step: env %d was never reset", i); return PMG_E_STATE; }
#pragma omp parallel for num_threads(e->nthreads) schedule(static)
    for (int i = 0; i < N; i++) {
        World* w = &e->w[i];
        env_step_one(e, w, actions + (size_t)i * dm->action_dim);
        float agl[20], dgl[20];
        env_obs(e, w, obs ? obs + (size_t)i * dm->observation_dim : NULL, pol ? pol + (size_t)i * dm->policy_state_dim : NULL, agl, dgl);
        /* reward from the double-precision goals (the reference's obs are float64) */
        double a64[20], d64[20];
        effective_goal(e, w, d64);
        if (e->cfg.task == PMG_TASK_REACH) {
            Kin k; kinematics(w->q, &k);
            for (int g = 0; g < 3; g++) a64[g] = k.p[PMG_BL_TIP][g];
        } else {
            int c0 = e->chest >= 0;
            if (c0) a64[0] = DOOR_Q(w);
            for (int b = 0; b < e->nb; b++)
                for (int g = 0; g < 3; g++) a64[c0 + 3 * b + g] = w->blk[b].pos[g];
            if (e->cfg.grip_informed_goal) {
                Kin k; kinematics(w->q, &k);
                real dd[3];
                v3sub(dd, k.p[PMG_BL_TAB1], k.p[PMG_BL_TAB2]);
                for (int g = 0; g < 3; g++) a64[c0 + 3 * e->nb + g] = k.p[PMG_BL_TIP][g];
                if (e->grasping) a64[c0 + 3 * e->nb + 3] = v3norm(dd);
            }
        }
        float r; uint8_t ok;
        reward_f64(e, a64, d64, dm->goal_dim, &r, &ok);
        if (ag) memcpy(ag + (size_t)i * dm->goal_dim, agl, sizeof(float) * dm->goal_dim);
        if (dg) memcpy(dg + (size_t)i * dm->goal_dim, dgl, sizeof(float) * dm->goal_dim);
        if (reward) reward[i] = r;
        if (goal_achieved) goal_achieved[i] = ok;
        if (done) done[i] = (uint8_t)(w->elapsed >= e->cfg.max_episode_steps);
    }
    return PMG_OK;
}

int pmgo_compute_reward(pmgo_env* e, const float* ag, const float* dg, int64_t batch, float* reward, uint8_t* ok)
{
    int G = e->dims.goal_dim;
    for (int64_t i = 0; i < batch; i++) {
        double a[20], d[20];
        for (int g = 0; g < G; g++) { a[g] = ag[i * G + g]; d[g] = dg[i * G + g]; }
        float r; uint8_t o;
        reward_f64(e, a, d, G, &r, &o);
        if (reward) reward[i] = r;
        if (ok) ok[i] = o;
    }
    return PMG_OK;
}

/* state layout (float32 per env), shared with the product (DESIGN.md "state row"):
 *  0-8 q | 9-17 qd | 18-20 ee_target | 21-27 joint_target | 28 grip_target | 29 elapsed |
 *  30 arm_enabled | 31 reset_count | 32-38 rest_pose | 39 - | 40-44 order | 45-47 base_target |
 *  48-62 desired_goal | 63 - | 64+13b: block b pos3 quat4 vel3 omg3 */
int pmgo_get_state(pmgo_env* e, float* state)
{
    int S = e->dims.state_dim;
    for (int i = 0; i < e->cfg.num_envs; i++) {
        const World* w = &e->w[i];
        float* s = state + (size_t)i * S;
        memset(s, 0, sizeof(float) * S);
        for (int d = 0; d < 9; d++) { s[d] = (float)w->q[d]; s[9 + d] = (float)w->qd[d]; }
        for (int a = 0; a < 3; a++) s[18 + a] = (float)w->ee_target[a];
        for (int d = 0; d < 7; d++) { s[21 + d] = (float)w->joint_target[d]; s[32 + d] = (float)w->rest_pose[d]; }
        s[28] = (float)w->grip_target; s[29] = (float)w->elapsed; s[30] = (float)w->arm_enabled; s[31] = (float)w->reset_count;
        for (int b = 0; b < NBMAX; b++) s[40 + b] = (float)w->order[b];
        for (int a = 0; a < 3; a++) s[45 + a] = (float)w->base_target[a];
        for (int g = 0; g < 15; g++) s[48 + g] = (float)w->goal[g];
        s[39] = (float)w->level; s[63] = (float)w->moved;
        if (e->cfg.use_curriculum) {
            float* cs = s + 64 + 13 * e->nb;
            int nc = num_curriculum(e), st = e->chest >= 0 ? 6 : 5;   /* prob | generated | goal_step */
            for (int b = 0; b < nc; b++) { cs[b] = (float)w->cur_prob[b]; cs[st + b] = (float)w->cur_count[b]; }
            cs[2 * st] = (float)w->cur_goal_step;
        }
        for (int b = 0; b < e->nb; b++) {
            const Block* bl = &w->blk[b];
            float* o = s + 64 + 13 * b;
            for (int a = 0; a < 3; a++) { o[a] = (float)bl->pos[a]; o[7 + a] = (float)bl->vel[a]; o[10 + a] = (float)bl->omg[a]; }
            for (int a = 0; a < 4; a++) o[3 + a] = (float)bl->quat[a];
        }
    }
    return PMG_OK;
}
int pmgo_set_state(pmgo_env* e, const float* state)
{
    int S = e->dims.state_dim;
    for (int i = 0; i < e->cfg.num_envs; i++) {
        World* w = &e->w[i];
        const float* s = state + (size_t)i * S;
        for (int d = 0; d < 9; d++) { w->q[d] = s[d]; w->qd[d] = s[9 + d]; }
        w->wsc_n = 0;
        for (int a = 0; a < 3; a++) w->ee_target[a] = s[18 + a];
        for (int d = 0; d < 7; d++) { w->joint_target[d] = s[21 + d]; w->rest_pose[d] = s[32 + d]; }
        w->grip_target = s[28]; w->elapsed = (int)s[29]; w->arm_enabled = (int)s[30]; w->reset_count = (int)s[31];
        for (int d = 7; d < 9; d++) { w->motor_target[d] = w->grip_target; w->motor_maximp[d] = FINGER_FORCE * PHYSICS_DT; }
        for (int b = 0; b < NBMAX; b++) w->order[b] = (int)s[40 + b];
        for (int a = 0; a < 3; a++) w->base_target[a] = s[45 + a];
        for (int g = 0; g < 15; g++) w->goal[g] = s[48 + g];
        w->level = (int)s[39]; w->moved = (int)s[63];
        if (e->cfg.use_curriculum) {
            const float* cs = s + 64 + 13 * e->nb;
            int nc = num_curriculum(e), st = e->chest >= 0 ? 6 : 5;
            for (int b = 0; b < nc; b++) { w->cur_prob[b] = cs[b]; w->cur_count[b] = cs[st + b]; }
            w->cur_goal_step = (int)cs[2 * st];
        }
        for (int b = 0; b < e->nb; b++) {
            Block* bl = &w->blk[b];
            const float* o = s + 64 + 13 * b;
            for (int a = 0; a < 3; a++) { bl->pos[a] = o[a]; bl->vel[a] = o[7 + a]; bl->omg[a] = o[10 + a]; }
            for (int a = 0; a < 4; a++) bl->quat[a] = o[3 + a];
        }
    }
    return PMG_OK;
}
int pmgo_set_goal(pmgo_env* e, const uint8_t* mask, const float* goals)
{
    int G = e->dims.goal_dim;
    if (e->chest >= 0) { snprintf(e->err, sizeof(e->err), "pmgo_set_goal: the chest tasks have no static target"); return PMG_E_INVALID; }
    for (int i = 0; i < e->cfg.num_envs; i++)
        if (!mask || mask[i])
            for (int g = 0; g < G && g < 15; g++) e->w[i].goal[g] = goals[(size_t)i * G + g]; /* static targets only */
    return PMG_OK;
}

/* set_sub_goal: kuka_multi_step_base_env.py:154-177 */
int pmgo_set_sub_goal(pmgo_env* e, const uint8_t* mask, int32_t ind)
{
    if (!e->cfg.task_decomposition) { snprintf(e->err, sizeof(e->err), "pmgo_set_sub_goal: task_decomposition is off"); return PMG_E_STATE; }
    int steps = e->chest >= 0 ? chest_num_steps(e) : (e->cfg.grip_informed_goal ? 2 * e->nb : e->nb); /* kuka_multi_step_envs.py:13-17 */
    if (ind < -1 || ind >= steps) { snprintf(e->err, sizeof(e->err), "pmgo_set_sub_goal: index %d out of range", ind); return PMG_E_INVALID; }
    for (int i = 0; i < e->cfg.num_envs; i++)
        if (!mask || mask[i]) e->w[i].level = ind < 0 ? steps - 1 : ind;
    return PMG_OK;
}
int pmgo_curriculum_update(pmgo_env* e, int32_t enabled)
{
    if (!e->cfg.use_curriculum) { snprintf(e->err, sizeof(e->err), "pmgo_curriculum_update: use_curriculum is off"); return PMG_E_STATE; }
    e->curriculum_update = enabled != 0;
    return PMG_OK;
}
int pmgo_curriculum_read(pmgo_env* e, int32_t* level, int32_t* goal_step, float* prob, float* generated)
{
    if (!e->cfg.use_curriculum) { snprintf(e->err, sizeof(e->err), "pmgo_curriculum_read: use_curriculum is off"); return PMG_E_STATE; }
    for (int i = 0; i < e->cfg.num_envs; i++) {
        const World* w = &e->w[i];
        if (level) level[i] = w->level;
        if (goal_step) goal_step[i] = w->cur_goal_step;
        int nc = num_curriculum(e);
        for (int b = 0; b < nc; b++) {
            if (prob) prob[(size_t)i * nc + b] = (float)w->cur_prob[b];
            if (generated) generated[(size_t)i * nc + b] = (float)w->cur_count[b];
        }
    }
    return PMG_OK;
}

/* ------------------------------------------------------------------ */
/* Bullet-call-level access to ONE world (env 0)                        */
/* ------------------------------------------------------------------ */
/* tools/refharness runs the REFERENCE's own Python (make_env, reset, step, _get_obs, _compute_reward, curricula,
 * sub-goals) on top of a scripted BulletClient whose physics calls land here: the orchestration is then the
 * reference's, the physics this file's, and tests/golden/ref_*.json records what the pair produces.  The env-level
 * entry points above (pmgo_reset / pmgo_step) must reproduce those records from the same seeds and actions -- which
 * pins every orchestration row of SURVEY.md section 8(a) to reference code.  The physics itself (a21) stays
 * [BULLET-PRIOR].  body: 0 = the Kuka (dof 0..8 = iiwa_joint_1..7, finger1, finger2), 1 = the chest (dof 0 = door). */
static World* bw_world(pmgo_env* e) { return &e->w[0]; }

/* resetJointState (robot_bases.py:192-199, 230-234) */
int pmgo_bw_reset_joint(pmgo_env* e, int body, int dof, double q, double qd)
{
    World* w = bw_world(e);
    if (body == 0 && dof >= 0 && dof < NJ) { w->q[dof] = (real)q; w->qd[dof] = (real)qd; return PMG_OK; }
    if (body == 1 && dof == 0 && e->chest >= 0) { DOOR_Q(w) = (real)q; DOOR_QD(w) = (real)qd; return PMG_OK; }
    return PMG_E_INVALID;
}
/* setJointMotorControl2 / Array in POSITION_CONTROL with gains 0.03 / 1.0 (kuka.py:282-301, chest.py:59-68);
 * force 0 is robot_bases.py:236-238's disable_motor() */
int pmgo_bw_motor(pmgo_env* e, int body, int dof, double target, double force)
{
    World* w = bw_world(e);
    if (body == 0 && dof >= 0 && dof < NJ) { w->motor_target[dof] = (real)target; w->motor_maximp[dof] = (real)force * PHYSICS_DT; return PMG_OK; }
    if (body == 1 && dof == 0 && e->chest >= 0) {
        /* the door motor of this restatement is a latch with the fixed target door_open and 500 N (chest.py:59-68) */
        if (force == 0) { DOOR_MOTOR(w) = 0; return PMG_OK; }
        if (force != 500.0 || (real)target != e->door_open) return PMG_E_INVALID;
        DOOR_MOTOR(w) = 1;
        return PMG_OK;
    }
    return PMG_E_INVALID;
}
int pmgo_bw_joint_state(pmgo_env* e, int body, int dof, double out[2])
{
    const World* w = bw_world(e);
    if (body == 0 && dof >= 0 && dof < NJ) { out[0] = w->q[dof]; out[1] = w->qd[dof]; return PMG_OK; }
    if (body == 1 && dof == 0 && e->chest >= 0) { out[0] = DOOR_Q(w); out[1] = DOOR_QD(w); return PMG_OK; }
    return PMG_E_INVALID;
}
/* getLinkState(computeLinkVelocity=1) of a robot link: COM position (= link frame origin for every link the
 * reference queries: their inertial origins are zero), orientation xyzw, world linear and angular velocity */
int pmgo_bw_link_state(pmgo_env* e, int link, double out[13])
{
    const World* w = bw_world(e);
    if (link < 0 || link >= NL) return PMG_E_INVALID;
    Kin k;
    kinematics(w->q, &k);
    real qt[4], v[3], om[3];
    R_to_quat(k.R[link], qt);
    point_velocity(&k, w->qd, link, k.c[link], v, om);
    for (int a = 0; a < 3; a++) { out[a] = k.c[link][a]; out[7 + a] = v[a]; out[10 + a] = om[a]; }
    for (int a = 0; a < 4; a++) out[3 + a] = qt[a];
    return PMG_OK;
}
/* resetBasePositionAndOrientation of free body b: velocities are zeroed */
int pmgo_bw_set_block(pmgo_env* e, int b, const double pos[3], const double quat[4])
{
    World* w = bw_world(e);
    if (b < 0 || b >= e->nb) return PMG_E_INVALID;
    Block* bl = &w->blk[b];
    for (int a = 0; a < 3; a++) { bl->pos[a] = (real)pos[a]; bl->vel[a] = 0; bl->omg[a] = 0; }
    for (int a = 0; a < 4; a++) bl->quat[a] = (real)quat[a];
    return PMG_OK;
}
int pmgo_bw_block_state(pmgo_env* e, int b, double out[13])
{
    const World* w = bw_world(e);
    if (b < 0 || b >= e->nb) return PMG_E_INVALID;
    const Block* bl = &w->blk[b];
    for (int a = 0; a < 3; a++) { out[a] = bl->pos[a]; out[7 + a] = bl->vel[a]; out[10 + a] = bl->omg[a]; }
    for (int a = 0; a < 4; a++) out[3 + a] = bl->quat[a];
    return PMG_OK;
}
/* calculateInverseKinematics from the current joint state (kuka.py:258-280) */
int pmgo_bw_ik(pmgo_env* e, const double pos[3], const double quat[4], int max_iter, double thr, double q_out[9])
{
    const World* w = bw_world(e);
    real t[3] = {(real)pos[0], (real)pos[1], (real)pos[2]}, tq[4] = {(real)quat[0], (real)quat[1], (real)quat[2], (real)quat[3]}, qo[NJ];
    int it = ik_solve(w->q, t, tq, max_iter, (real)thr, qo);
    for (int d = 0; d < NJ; d++) q_out[d] = qo[d];
    return it;
}
/* stepSimulation: 20 substeps of 2 ms (base_env.py:203-220) */
int pmgo_bw_step_simulation(pmgo_env* e)
{
    step_simulation(e, bw_world(e));
    return PMG_OK;
}

/* ---- [BULLET-PRIOR] switches (see the table at the top) ---- */
int pmgo_prior_count(void) { return PRIOR_N; }
const char* pmgo_prior_name(int i) { return i >= 0 && i < PRIOR_N ? PRIOR_NAME[i] : NULL; }
int pmgo_set_prior(const char* name, double value)
{
    for (int i = 0; i < PRIOR_N; i++)
        if (strcmp(name, PRIOR_NAME[i]) == 0) { G_PRIOR[i] = value; return PMG_OK; }
    return PMG_E_INVALID;
}
double pmgo_get_prior(const char* name)
{
    for (int i = 0; i < PRIOR_N; i++)
        if (strcmp(name, PRIOR_NAME[i]) == 0) return G_PRIOR[i];
    return -1;
}
void pmgo_reset_priors(void) { memcpy(G_PRIOR, PRIOR_DEFAULT, sizeof(G_PRIOR)); }
