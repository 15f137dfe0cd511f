/*
 * pmg_oracle.h -- CPU restatement (TEST INFRASTRUCTURE, not product code).
 *
 * PARITY UNPINNED: the arithmetic of this hot path lives in the third-party
 * dependency pybullet~=3.0.6 (reference requirements.txt:2), which is absent
 * from the build container and from the GPU box, and the reference's own
 * tests hold no assertions or golden vectors.  The
 * physics below restates Bullet 3.0.x's published multibody algorithm
 * ([BULLET-PRIOR] in the comments); the parts that CAN be pinned (RNG stream,
 * sampling rules, observation layout, reward, FK known-answer) are pinned by
 * tests/golden/.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (pybullet_multigoal_gym_amd) never does.
 *
 * The ABI mirrors include/pmg.h with a pmgo_ prefix; I/O is float32 like the
 * product, internal arithmetic is double (or float with -DPMGO_FLOAT).
 */
#ifndef PMG_ORACLE_H
#define PMG_ORACLE_H
#include <stdint.h>
#include "../include/pmg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmgo_env pmgo_env;

int pmgo_create(const pmg_config* cfg, pmgo_env** out);
void pmgo_destroy(pmgo_env* env);
int pmgo_get_dims(const pmgo_env* env, pmg_dims* out);
const char* pmgo_last_error(const pmgo_env* env);
int pmgo_seed(pmgo_env* env, uint64_t seed_base, uint64_t seed_stride);
int pmgo_reset(pmgo_env* env, const uint8_t* mask, float* observation, float* policy_state,
               float* achieved_goal, float* desired_goal);
int pmgo_step(pmgo_env* env, const float* actions, float* observation, float* policy_state,
              float* achieved_goal, float* desired_goal, float* reward, uint8_t* goal_achieved,
              uint8_t* done);
int pmgo_compute_reward(pmgo_env* env, const float* achieved_goal, const float* desired_goal,
                        int64_t batch, float* reward, uint8_t* goal_achieved);
int pmgo_get_state(pmgo_env* env, float* state);
int pmgo_set_state(pmgo_env* env, const float* state);
int pmgo_set_goal(pmgo_env* env, const uint8_t* mask, const float* goals);
int pmgo_set_sub_goal(pmgo_env* env, const uint8_t* mask, int32_t sub_goal_ind);
int pmgo_curriculum_update(pmgo_env* env, int32_t enabled);
int pmgo_curriculum_read(pmgo_env* env, int32_t* level, int32_t* goal_step, float* prob, float* generated);
int pmgo_set_threads(pmgo_env* env, int nthreads);   /* OpenMP threads for the env loop */

/* ---- probes used by the unit tests (double precision, single env) ---- */
/* forward kinematics of the 9-dof chain: tip position, rotation (row-major 3x3) */
void pmgo_fk_tip(const double q[9], double pos[3], double rot[9]);
/* restated calculateInverseKinematics (kuka.py:258-280): returns iterations used */
int pmgo_ik(const double q_start[9], const double target_pos[3], const double target_quat_xyzw[4],
            int max_iter, double threshold, double q_out[9]);
/* joint-space inverse mass matrix via the ABA impulse response (81 doubles) */
void pmgo_minv(const double q[9], double minv[81]);
/* forward dynamics qdd(q, qd, tau) with gravity and Bullet's link damping */
void pmgo_fdyn(const double q[9], const double qd[9], const double tau[9], double qdd[9]);
/* box-box narrowphase probe: returns n contacts (<=4); out[n][10] = pa3 pb3 n3 dist */
int pmgo_box_box(const double ca[3], const double Ra[9], const double ha[3], const double cb[3],
                 const double Rb[9], const double hb[3], double margin, double* out);
/* cylinder (A) x box (B) narrowphase probe, same output convention */
int pmgo_cyl_box(const double cc[3], const double Rc[9], double rad, double hl, const double cb[3], const double Rb[9],
                 const double hb[3], double margin, double* out);
/* gym.utils.seeding.np_random(seed) + RandomState draws (tests/golden/rng.json) */
void pmgo_rng_probe(uint64_t seed, int n_double, double* out_double, int shuffle_n, int32_t* out_perm);


/* ---- Bullet-call-level access to one world (env 0), used by tools/refharness to run the reference's own Python on
 * this file's physics (see the comment in pmg_oracle.c); body 0 = robot (dof 0..8), 1 = chest (dof 0 = door) ---- */
int pmgo_bw_reset_joint(pmgo_env* env, int body, int dof, double q, double qd);
int pmgo_bw_motor(pmgo_env* env, int body, int dof, double target, double force);
int pmgo_bw_joint_state(pmgo_env* env, int body, int dof, double out[2]);
int pmgo_bw_link_state(pmgo_env* env, int bullet_link, double out[13]);   /* pos3 quat_xyzw4 lin3 ang3 */
int pmgo_bw_set_block(pmgo_env* env, int b, const double pos[3], const double quat_xyzw[4]);
int pmgo_bw_block_state(pmgo_env* env, int b, double out[13]);
int pmgo_bw_ik(pmgo_env* env, const double pos[3], const double quat_xyzw[4], int max_iter, double threshold, double q_out[9]);
int pmgo_bw_step_simulation(pmgo_env* env);


/* ---- [BULLET-PRIOR] switches: process-wide alternatives for the choices this restatement had to make without PyBullet
 * (names and defaults: the table at the top of pmg_oracle.c).  Set before stepping; pmgo_reset_priors() restores the
 * defaults, which are what the product compiles in. ---- */
int pmgo_prior_count(void);
const char* pmgo_prior_name(int i);
int pmgo_set_prior(const char* name, double value);
double pmgo_get_prior(const char* name);
void pmgo_reset_priors(void);

#ifdef __cplusplus
}
#endif
#endif
