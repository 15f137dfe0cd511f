/* TEST INFRASTRUCTURE: driver of the AddressSanitizer / UndefinedBehaviorSanitizer build of the oracle
 * (tests/test_sanitizers.py).  Every task, with its options, through create / reset / masked
 * reset / random steps / compute_reward / get_state / set_state / destroy. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pmg_oracle.h"

static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void)
{
    const int N = 6, T = 12;
    int rc_all = 0;
    for (int task = 0; task < 8; task++)
        for (int variant = 0; variant < 2; variant++) {
            pmg_config c;
            memset(&c, 0, sizeof(c));
            c.struct_size = (int)sizeof(c);
            c.task = task; c.num_envs = N; c.num_block = task >= 4 ? (variant ? 5 : 2) : 1;
            c.binary_reward = !variant; c.max_episode_steps = 5; c.distance_threshold = 0.05f; c.random_order = 1;
            c.seed_base = 7; c.seed_stride = 1;
            c.joint_control = variant && task < 4;
            c.use_curriculum = variant && task >= 4;
            c.num_goals_to_generate = 40;
            c.task_decomposition = !variant && (task == 4 || task >= 6);
            c.grip_informed_goal = variant && (task == 4 || task >= 6) ? 0 : (!variant && task == 4);
            pmgo_env* e = NULL;
            if (pmgo_create(&c, &e) != 0) { fprintf(stderr, "create failed: task %d variant %d: %s\n", task, variant, pmgo_last_error(NULL)); return 2; }
            pmg_dims d;
            pmgo_get_dims(e, &d);
            pmgo_set_threads(e, 2);
            float* obs = malloc(sizeof(float) * N * d.observation_dim);
            float* pol = malloc(sizeof(float) * N * d.policy_state_dim);
            float* ag = malloc(sizeof(float) * N * d.goal_dim);
            float* dg = malloc(sizeof(float) * N * d.goal_dim);
            float* act = malloc(sizeof(float) * N * d.action_dim);
            float* st = malloc(sizeof(float) * N * d.state_dim);
            float r[6]; uint8_t ok[6], dn[6], mask[6] = {1, 0, 1, 0, 1, 0};
            unsigned s = 12345u + task;
            rc_all |= pmgo_reset(e, NULL, obs, pol, ag, dg);
            if (c.use_curriculum) pmgo_curriculum_update(e, 1);
            for (int t = 0; t < T; t++) {
                for (int i = 0; i < N * d.action_dim; i++) act[i] = (float)(lcg(&s) % 2001) / 1000.f - 1.f;
                rc_all |= pmgo_step(e, act, obs, pol, ag, dg, r, ok, dn);
                rc_all |= pmgo_compute_reward(e, ag, dg, N, r, ok);
                if (t % 5 == 4) rc_all |= pmgo_reset(e, mask, obs, pol, ag, dg);
                if (c.task_decomposition && t == 3) pmgo_set_sub_goal(e, NULL, 0);
            }
            pmgo_get_state(e, st);
            pmgo_set_state(e, st);
            rc_all |= pmgo_step(e, act, obs, pol, ag, dg, r, ok, dn);
            free(obs); free(pol); free(ag); free(dg); free(act); free(st);
            pmgo_destroy(e);
        }
    printf("oracle sanitizer run: 8 tasks x 2 variants, rc %d\n", rc_all);
    return rc_all != 0;
}
