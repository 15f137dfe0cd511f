#!/usr/bin/env python3
"""Task-level sensitivity of the CPU oracle to every switchable [BULLET-PRIOR] (oracle/pmg_oracle.c: the choices the
restatement had to make without PyBullet) -- what an RL user would inherit from each of them.  CPU only.

    tools/prior_sensitivity.py [N_scripted] [N_random] [threads] > profiles/r04_prior_sensitivity.jsonl

For the default setting and for every single-switch alternative:
  * the scripted task-solving suite (tools/scripted_suite.py: reach, pick_and_place, push, slide, block_stack-2/-4,
    block_rearrange-2, chest_push, chest_pick_and_place): success (ever / at the last step), N_scripted envs per task;
  * a random-policy rollout of N_random envs x 50 steps for push, pick_and_place, block_stack-4 and chest_push-4 with the
    SAME seeds and actions as the default run: percentiles of |achieved_goal - achieved_goal(default)| at the last step,
    of the object displacement from its start, the fraction of goals achieved, how many blocks left the table, and the
    per-column mean / standard deviation of the observation over the whole rollout next to the default's.
One JSON line per (alternative, task, kind).  An alternative that moves a scripted success rate by more than 5 % is what a
first real PyBullet capture (tools/gen_reference_fixtures.py --real) has to settle first."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import oracle_lib  # noqa: E402
import scripted_policies as SP  # noqa: E402
import scripted_suite as SS  # noqa: E402

# (prior, value): one switch at a time against the defaults (the table at the top of oracle/pmg_oracle.c)
ALTERNATIVES = [
    ('default', None),
    ('contact_warm_start', 0.85),      # btSequentialImpulseConstraintSolver's factor; btMultiBody compiles it out
    ('solver_iterations', 10.0),       # the reference sets numSolverIterations=5 (base_env.py:37); Bullet's default is 50
    ('solver_iterations', 50.0),
    ('residual_threshold', 0.0),       # no early exit of the Gauss-Seidel sweeps
    ('friction_dirs', 1.0),            # one friction direction (along the sliding velocity) instead of two btPlaneSpace1 axes
    ('motor_impulse_dt', 0.002),       # motor clamp force x substep instead of force x physicsDeltaTime
    ('link_damping', 0.0),             # no multibody link damping
    ('damping_per_substep', 1.0),      # joint damping re-evaluated every substep instead of latched per stepSimulation
    ('warm_start', 0.85),              # warm starting of the non-contact rows
    ('contact_margin', 0.0),           # contacts only when penetrating
    ('contact_margin', 0.02),          # Bullet's 2 cm manifold breaking threshold as a speculative margin
    ('joint_erp', 0.8),
    ('linear_slop', 0.0),
    ('ik_damping', 0.1),
]
RANDOM_TASKS = [('push', {}), ('pick_and_place', {}), ('block_stack', {'num_block': 4}), ('chest_push', {'num_block': 4})]


def scripted(name, N, threads):
    task = name.rsplit('_', 1)[0] if name.startswith(('block_stack_', 'block_rearrange_')) else name
    kw, T = SS.SUITE[name]
    env = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=threads, max_episode_steps=T, **kw)
    env.reset()
    obs = env.reset()
    pol = SP.make_policy(task, N, **({'num_block': kw['num_block']} if 'num_block' in kw else {}))
    obs, ok, ever = SP.rollout(env, pol, T, obs)
    env.close()
    return ok, ever


def random_rollout(task, kw, N, T, threads):
    env = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=threads, **kw)
    env.reset()
    o0 = env.reset()
    rs = np.random.RandomState(12345)
    A = env.dims.action_dim
    s1 = np.zeros(env.dims.observation_dim)
    s2 = np.zeros(env.dims.observation_dim)
    achieved = np.zeros(N, bool)
    for t in range(T):
        o, r, d, ok = env.step(rs.uniform(-1, 1, (N, A)).astype(np.float32))
        achieved |= ok
        x = o['observation'].astype(np.float64)
        s1 += x.sum(0)
        s2 += (x * x).sum(0)
    env.close()
    n = N * T
    mean = s1 / n
    std = np.sqrt(np.maximum(s2 / n - mean * mean, 0.0))
    return o0['achieved_goal'].copy(), o['achieved_goal'].copy(), achieved, mean, std


def main():
    Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    Nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else oracle_lib.usable_threads()
    base = {}
    warnings.simplefilter('ignore')
    for prior, value in ALTERNATIVES:
        oracle_lib.reset_priors()
        if value is not None:
            oracle_lib.set_prior(prior, value)
        tag = prior if value is None else '%s=%g' % (prior, value)
        t0 = time.time()
        for name in SS.SUITE:
            ok, ever = scripted(name, Ns, threads)
            row = {'alternative': tag, 'kind': 'scripted', 'task': name, 'N': Ns, 'T': SS.SUITE[name][1],
                   'success_ever': float(ever.mean()), 'success_end': float(ok.mean())}
            if value is None:
                base[('s', name)] = ever.copy()
            else:
                b = base[('s', name)]
                row['delta_success_ever'] = float(ever.mean() - b.mean())
                row['envs_with_a_different_outcome'] = int((ever != b).sum())
            print(json.dumps(row), flush=True)
        for task, kw in RANDOM_TASKS:
            ag0, ag, achieved, mean, std = random_rollout(task, kw, Nr, 50, threads)
            c0 = 1 if task.startswith('chest') else 0
            nb = kw.get('num_block', 1)
            z = ag[:, c0:c0 + 3 * nb].reshape(Nr, -1, 3)[..., 2]
            moved = np.abs(ag - ag0).max(1)
            row = {'alternative': tag, 'kind': 'random', 'task': task + ('-%d' % nb if nb > 1 else ''), 'N': Nr, 'T': 50,
                   'goal_achieved_ever': float(achieved.mean()), 'blocks_off_the_table': int((z < 0.1).sum()),
                   'object_displacement_p50_p90_p99': [float(np.percentile(moved, q)) for q in (50, 90, 99)]}
            if value is None:
                base[('r', task)] = (ag.copy(), mean.copy(), std.copy(), achieved.copy())
            else:
                bag, bmean, bstd, bach = base[('r', task)]
                err = np.abs(ag - bag).max(1)
                row['final_goal_vs_default_p50_p90_p99'] = [float(np.percentile(err, q)) for q in (50, 90, 99)]
                row['envs_beyond_1e-3'] = float((err > 1e-3).mean())
                row['envs_beyond_1e-2'] = float((err > 1e-2).mean())
                scale = np.maximum(bstd, 1e-6)
                row['obs_mean_shift_max_in_default_std'] = float((np.abs(mean - bmean) / scale).max())
                row['obs_std_ratio_min_max'] = [float((std / scale).min()), float((std / scale).max())]
                row['delta_goal_achieved_ever'] = float(achieved.mean() - bach.mean())
            print(json.dumps(row), flush=True)
        print('%s: %.0f s' % (tag, time.time() - t0), file=sys.stderr, flush=True)
    oracle_lib.reset_priors()


if __name__ == '__main__':
    main()
