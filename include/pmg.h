/*
 * pmg.h -- C ABI of the MI355X-native vectorised multigoal manipulation env.
 *
 * This is the drop-in boundary for the reference's per-step hot path.  The
 * reference (pure Python) crosses into native code through ~25 PyBullet C-API
 * entry points per env step; this library replaces
 * that whole inner boundary with ONE batched call per phase.  Each entry
 * point below cites the reference interface it replaces
 * (P/ = pybullet_multigoal_gym/ in the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / numpy types.
 *   - Every function returns 0 on success, a negative PMG_E_* code otherwise;
 *     the message is available from pmg_last_error().  Nothing aborts or
 *     throws across this boundary.
 *   - The caller owns every host buffer passed in; the library owns the
 *     handle and all device memory.  Buffers named d_* are DEVICE pointers
 *     (HIP) supplied by the caller (e.g. a torch tensor's data_ptr()).
 *   - A handle is used from one host thread at a time.  Host-buffer calls
 *     are synchronous at return; *_device calls are stream-ordered on the
 *     handle's HIP stream (pmg_sync() waits for it).
 *   - All arrays are row-major with a leading num_envs axis, float32.
 */
#ifndef PMG_H
#define PMG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tasks: P/__init__.py:14-44 ('reach','push','pick_and_place','slide','block_stack','block_rearrange','chest_push',
 * 'chest_pick_and_place') */
enum {
    PMG_TASK_REACH = 0,
    PMG_TASK_PUSH = 1,
    PMG_TASK_PICK_AND_PLACE = 2,
    PMG_TASK_SLIDE = 3,
    PMG_TASK_BLOCK_STACK = 4,
    PMG_TASK_BLOCK_REARRANGE = 5,
    PMG_TASK_CHEST_PUSH = 6,            /* KukaChestPushEnv: front sliding door (kuka_multi_step_envs.py:385-403) */
    PMG_TASK_CHEST_PICK_AND_PLACE = 7   /* KukaChestPickAndPlaceEnv: up sliding lid (:230-254) */
};

enum {
    PMG_OK = 0,
    PMG_E_INVALID = -1,     /* bad argument / unsupported configuration */
    PMG_E_DEVICE = -2,      /* HIP runtime error */
    PMG_E_NOMEM = -3,
    PMG_E_STATE = -4,       /* call order error (e.g. step before reset) */
    PMG_E_COMM = -5         /* RCCL error */
};

/* which device buffer pmg_device_ptr() returns */
enum {
    PMG_BUF_PACKED = 7,  /* [N, packed_dim] float32 rows: observation | policy_state | achieved_goal |
                            desired_goal | reward | goal_achieved (0/1) | done (0/1); widths in pmg_dims */
    PMG_BUF_STATE = 8,   /* [N, 32] float32 hot state rows (q9 qd9 ee3 jt7 grip elapsed enabled resets) */
    PMG_BUF_SCHED = 9,   /* [4 + 3N + 3 ceil(N / 1024)] int32 launch schedule of the last step (diagnostics): n_prone, n_free,
                            prone list [N], free list [N], n_redo, redo list [N], then the two-pass plan's per-workgroup class
                            counts and its promotion flag -- see DESIGN.md "launch-order plan" */
    PMG_BUF_ENV_CYCLES = 10 /* [N, 2] int32: shader cycles / 64 the env's wavefront spent in its last step, and the largest contact
                            count any of that step's substeps saw (envs with free objects).  Kept when the handle was created with
                            PMG_ENV_CYCLES=1 in the environment (diagnostics) AND whenever the longest-first order of the fast-path
                            list is on (block_stack and the chest tasks from 4096 envs, PMG_LPT_CYCLES; the plan's predictor: 8 B per
                            env written every step); PMG_E_INVALID otherwise */
};

/* POD configuration; mirrors the kwargs of pmg.make_env (P/__init__.py:4-11)
 * that the hot path honours, plus the batch geometry.  Zero-initialise and
 * set struct_size = sizeof(pmg_config). */
typedef struct pmg_config {
    int32_t struct_size;
    int32_t task;               /* PMG_TASK_* */
    int32_t num_envs;           /* N: envs simulated by THIS handle (one GPU) */
    int32_t num_block;          /* multi-block tasks (block_stack, block_rearrange, chest_*), 1..5 (P/__init__.py:108) */
    int32_t binary_reward;      /* P/__init__.py:4 */
    int32_t joint_control;      /* P/__init__.py:6 */
    int32_t max_episode_steps;  /* gym TimeLimit, P/__init__.py:6,105 */
    int32_t device;             /* HIP device ordinal */
    float distance_threshold;   /* P/__init__.py:6 */
    int32_t random_order;       /* block_stack, kuka_multi_step_envs.py:7 */
    uint64_t seed_base;         /* env i is seeded with seed_base + i*seed_stride */
    uint64_t seed_stride;       /* 0 reproduces the reference (every env seed 0) */
    int32_t env_index_offset;   /* global index of this shard's env 0 (multi-GPU) */
    int32_t task_decomposition; /* block_stack, chest_*: sub-goals, kuka_multi_step_envs.py:89-122, 285-342, 433-475
                                   (excludes use_curriculum) */
    int32_t use_curriculum;     /* kuka_multi_step_base_env.py:121-140: num_block levels (block_stack / block_rearrange,
                                   num_block >= 2) or num_block + 1 (chest_*: how many blocks go into the chest) */
    int32_t num_goals_to_generate; /* curriculum budget, P/__init__.py:11 (default 1e6); 0 = 1e6 */
    int32_t grip_informed_goal; /* block_stack: goals carry the gripper tip target + finger width (kuka_multi_step_envs.py:75-77);
                                   goal_dim = 3*num_block + 4, sub-goals double (pick / place).  chest_*: goal_dim =
                                   1 + 3*num_block (door joint first) + 3 (chest_push: tip) / + 4 (chest_pick_and_place: tip,
                                   finger width); 2 / 3 sub-goals per block after "open the door" */
    int32_t reserved[3];
} pmg_config;

typedef struct pmg_dims {
    int32_t num_envs;
    int32_t action_dim;      /* kuka.py:103-118 */
    int32_t observation_dim; /* kuka_single_step_base_env.py:193-221; kuka_multi_step_base_env.py:255-336 */
    int32_t policy_state_dim;
    int32_t goal_dim;        /* achieved_goal and desired_goal */
    int32_t state_dim;       /* floats per env in get_state/set_state */
    int32_t packed_dim;      /* floats per env in PMG_BUF_PACKED */
    int32_t reserved;
} pmg_dims;

typedef struct pmg_env pmg_env;

/* Replaces: pmg.make_env -> gym.make -> TaskEnv.__init__ -> BaseBulletMGEnv.__init__
 * (P/__init__.py:178; base_env.py:15-110): creates N worlds with the physics
 * parameters of base_env.py:203-220 and seeds them.  Does NOT perform the
 * constructor's implicit reset (base_env.py:84); the host layer does. */
int pmg_create(const pmg_config* cfg, pmg_env** out);
void pmg_destroy(pmg_env* env);
int pmg_get_dims(const pmg_env* env, pmg_dims* out);
const char* pmg_last_error(const pmg_env* env); /* env may be NULL: last create error */
/* HIP devices this process can see (0 when there is none / no driver): lets a rank whose launcher restricted
 * visibility to one GPU fall back to device 0 instead of LOCAL_RANK (bench.py). */
int pmg_device_count(void);

/* Replaces: BaseBulletMGEnv.seed (base_env.py:120-122) == gym.utils.seeding.np_random:
 * MT19937 seeded by init_by_array(sha512(str(seed))[:8]) per env. */
int pmg_seed(pmg_env* env, uint64_t seed_base, uint64_t seed_stride);

/* Replaces: BaseBulletMGEnv.reset (base_env.py:124-128) = Kuka.robot_specific_reset
 * (kuka.py:120-165) + _task_reset/_generate_goal (kuka_single_step_base_env.py:76-148;
 * kuka_multi_step_base_env.py:183-250; kuka_multi_step_envs.py:34-87) + _get_obs.
 * mask: N bytes (nonzero = reset this env) or NULL = all.  Output pointers may
 * be NULL to skip the copy-out.  Envs outside the mask keep their state; their rows of PMG_BUF_PACKED get a fresh
 * observation / desired goal of that state and KEEP reward | goal_achieved | done of their last step (a device-resident
 * loop may step, reset(mask = done) and then still read every env's last reward). */
int pmg_reset(pmg_env* env, const uint8_t* mask, float* observation, float* policy_state,
              float* achieved_goal, float* desired_goal);

/* Replaces: TimeLimit.step -> BaseBulletMGEnv.step (base_env.py:130-138) =
 * Kuka.apply_action (kuka.py:167-225: tip-delta, clip, IK, motors,
 * 5 x stepSimulation) + _get_obs + _compute_reward.  actions: [N, action_dim]. */
int pmg_step(pmg_env* env, const float* actions, float* observation, float* policy_state,
             float* achieved_goal, float* desired_goal, float* reward, uint8_t* goal_achieved,
             uint8_t* done);

/* Device-resident variants: inputs already in HBM, outputs stay in the
 * library's device buffers (pmg_device_ptr).  Stream-ordered, no host sync. */
int pmg_reset_device(pmg_env* env, const uint8_t* d_mask);
/* The vectorised-env policy the reference leaves to its caller (one gym env is reset by whoever reads `done`): reset, on
 * the device and without a host mask, exactly the envs whose episode has ended -- TimeLimit: elapsed steps >=
 * max_episode_steps (gym TimeLimit.step; reference: P/__init__.py:148-177 `max_episode_steps`).  Same state as
 * pmg_reset_device with the mask of those envs; in the packed row the observation / goals are the new episode's first
 * ones while reward | goal_achieved | done keep the finished step's values (the usual auto-reset convention).
 * Idempotent: a freshly reset env has elapsed = 0. */
int pmg_reset_done_device(pmg_env* env);
int pmg_step_device(pmg_env* env, const float* d_actions);
int pmg_device_ptr(pmg_env* env, int which, void** d_ptr);
int pmg_stream(pmg_env* env, void** hip_stream);
int pmg_sync(pmg_env* env);
/* copy the current output buffers to host (any pointer may be NULL) */
int pmg_read_outputs(pmg_env* env, float* observation, float* policy_state, float* achieved_goal,
                     float* desired_goal, float* reward, uint8_t* goal_achieved, uint8_t* done);

/* Replaces: KukaBulletMGEnv._compute_reward (kuka_single_step_base_env.py:237-244;
 * kuka_multi_step_base_env.py:338-345) on [B, goal_dim] batches (HER relabelling). */
int pmg_compute_reward(pmg_env* env, const float* achieved_goal, const float* desired_goal,
                       int64_t batch, float* reward, uint8_t* goal_achieved);
int pmg_compute_reward_device(pmg_env* env, const float* d_achieved_goal, const float* d_desired_goal,
                              int64_t batch, float* d_reward, uint8_t* d_goal_achieved);

/* Checkpoint / test hooks (no reference equivalent).
 * state: [N, state_dim] float32, layout documented in DESIGN.md (with use_curriculum the row ends with 16
 * floats of curriculum state: prob[5] generated[5] goal_step; chest tasks prob[6] generated[6] goal_step). */
int pmg_get_state(pmg_env* env, float* state);
int pmg_set_state(pmg_env* env, const float* state);   /* also refreshes the observation part of PMG_BUF_PACKED */
/* Host-injected goal / object poses for seed-parity tests (replaces the RNG
 * draws of _generate_goal for the masked envs).  goals: [N, goal_dim].  PMG_E_INVALID for the chest tasks (their goal
 * is the chest: no static target; goal row [0..2] holds the door joint position, velocity and motor latch). */
int pmg_set_goal(pmg_env* env, const uint8_t* mask, const float* goals);

/* Multi-step task bookkeeping (block_stack / block_rearrange), per env -- the reference keeps one copy per
 * env object.  The desired goal of these tasks is re-derived from the current block poses at every observation
 * (kuka_multi_step_base_env.py:309-312): blocks beyond the active level are "already at their goal".
 *
 * Replaces: KukaBulletMultiBlockEnv.set_sub_goal (kuka_multi_step_base_env.py:154-177): sub-goal index for
 * the masked envs (mask NULL = all), -1 = the final goal, as after reset; refreshes desired_goal in the output
 * buffers.  Valid indices: [-1, num_block), or [-1, 2*num_block) with grip_informed_goal (pick, place, pick, ...);
 * chest tasks: [-1, num_steps) with num_steps = num_block + 1, or 2*num_block + 1 (chest_push) / 3*num_block + 1
 * (chest_pick_and_place) with grip_informed_goal -- index 0 is "open the door" (kuka_multi_step_envs.py:238-242,
 * 388-392).  PMG_E_STATE unless the handle was created with task_decomposition. */
int pmg_set_sub_goal(pmg_env* env, const uint8_t* mask, int32_t sub_goal_ind);
/* Replaces: activate_curriculum_update / deactivate_curriculum_update (kuka_multi_step_base_env.py:142-152). */
int pmg_curriculum_update(pmg_env* env, int32_t enabled);
/* Curriculum read-out, any pointer may be NULL: level [N] (last_curriculum_level), goal_step [N]
 * (curriculum_goal_step = level*25 + 50), prob [N, num_curriculum] (curriculum_prob), generated [N, num_curriculum]
 * (num_generated_goals_per_curriculum); num_curriculum = num_block, or num_block + 1 for the chest tasks. */
int pmg_curriculum_read(pmg_env* env, int32_t* level, int32_t* goal_step, float* prob, float* generated);

/* Multi-GPU (no reference equivalent): one handle per
 * rank; the only exchange is an RCCL all-gather of PMG_BUF_PACKED. */
int pmg_comm_unique_id(uint8_t id[128]);
int pmg_comm_init(pmg_env* env, int rank, int nranks, const uint8_t id[128]);
/* d_gathered: [nranks*N, packed_dim] device buffer (caller-owned). */
int pmg_allgather_packed(pmg_env* env, float* d_gathered);
/* The same, OVERLAPPED with the next step (the collective must not add to the step).  pmg_comm_overlap(env, 1)
 * double-buffers the packed rows: step t writes buffer t & 1 (PMG_BUF_PACKED / pmg_device_ptr then names the buffer of the
 * LAST step: query it per step, or read the gathered rows).  pmg_allgather_packed_async enqueues ncclAllGather on the handle's own
 * communication stream behind the rows of the last step (+ the masked resets enqueued since); the step stream does not wait
 * for it -- only the step that writes the same row buffer again (two steps later) does.  d_gathered must stay untouched
 * until pmg_allgather_wait: host != 0 blocks the caller until the LAST enqueued all-gather has completed, host == 0 makes the
 * handle's stream wait for it (for consumers enqueued on pmg_stream()).  A caller that consumes gather t while gather t + 1 is in
 * flight alternates two d_gathered buffers.  pmg_sync() also waits for the communication stream. */
int pmg_comm_overlap(pmg_env* env, int32_t enabled);
int pmg_allgather_packed_async(pmg_env* env, float* d_gathered);
int pmg_allgather_wait(pmg_env* env, int32_t host);

/* Device-buffer helpers for callers that have no HIP runtime of their own (e.g. a numpy-only
 * host): allocate / free / copy on the handle's device and stream.  Callers that already own
 * device memory (a torch tensor's data_ptr()) pass those pointers directly instead. */
int pmg_device_alloc(pmg_env* env, uint64_t bytes, void** d_ptr);
int pmg_device_free(pmg_env* env, void* d_ptr);
int pmg_upload(pmg_env* env, void* d_dst, const void* h_src, uint64_t bytes);   /* synchronous at return */
int pmg_download(pmg_env* env, void* h_dst, const void* d_src, uint64_t bytes); /* synchronous at return */

/* Kernel timing of the most recent *_device call sequence: HIP events on the
 * handle's stream bracket every step kernel; returns average ms per launch
 * over the launches since the last pmg_timing_reset(). */
int pmg_timing_reset(pmg_env* env);
/* bracket only every n-th batched step with events (default 1: every step).  An event is a barrier packet with a
 * completion signal: the kernel behind it starts ~6 us after the one in front has drained (measured, rocprofv3 kernel
 * trace), against ~0.1 us between kernels that follow each other directly -- two events per step were 12 us of a 1.0 ms
 * reach step.  The untimed steps run the identical launch sequence. */
int pmg_timing_every(pmg_env* env, int n);
int pmg_timing_read(pmg_env* env, double* avg_step_kernel_ms, int64_t* launches);
/* the same launches: shortest / average / longest (any pointer may be NULL).  A batched step lasts as long as its slowest
 * wavefront, so the spread shows how often envs with finger x table / object contacts were in the batch. */
int pmg_timing_stats(pmg_env* env, double* min_ms, double* avg_ms, double* max_ms, int64_t* launches);
/* HIP events around every pmg_allgather_packed since the last pmg_timing_reset(): average / longest ms on this rank's
 * stream (it includes the wait for the slowest peer to arrive) and the count (any pointer may be NULL). */
int pmg_comm_timing(pmg_env* env, double* avg_ms, double* max_ms, int64_t* launches);

/* The per-env MT19937 streams (625 words each: state + cursor), [N, 625] uint32: with pmg_get_state / pmg_set_state a
 * checkpoint that resumes with the SAME future goals, orders and curriculum draws (no reference equivalent).  The
 * curriculum-update switch is host state: restore it with pmg_curriculum_update. */
int pmg_get_rng(pmg_env* env, uint32_t* words);
int pmg_set_rng(pmg_env* env, const uint32_t* words);

#ifdef __cplusplus
}
#endif
#endif /* PMG_H */
