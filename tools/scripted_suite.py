#!/usr/bin/env python3
"""Solvability of the tasks under the scripted controllers, free-running (no re-synchronisation), on the HIP library
and / or the CPU oracle:   tools/scripted_suite.py [device|oracle|both] [N]
Prints one JSON line per (task, backend): success at the last step, ever, and -- on the device -- how many env-steps
ran on the one-env-per-wavefront launch lists (the gripper-on-object kernels)."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import scripted_policies as SP  # noqa: E402

# task -> (make_env kwargs, episode length).  Episode lengths are make_env's max_episode_steps (P/__init__.py:6): the
# multi-step tasks need several pick-and-place cycles of ~50 steps each.
SUITE = {
    'reach': ({}, 50),
    'pick_and_place': ({}, 50),
    'push': ({}, 300),
    'slide': ({}, 60),
    'block_stack_2': ({'num_block': 2}, 130),
    'block_stack_4': ({'num_block': 4}, 340),
    'block_rearrange_2': ({'num_block': 2}, 400),
    'chest_push': ({'num_block': 1}, 360),
    'chest_pick_and_place': ({'num_block': 1}, 100),
}


def feasible_mask(task, obs0):
    """Envs whose object starts where the arm can get behind it: beyond x = -0.45 the tip (clip box upper x = -0.37,
    kuka.py:41; measured reach of the DLS IK there: -0.395 at y = 0) cannot be placed on the far side of a block, so a
    push towards the robot is kinematically impossible.  Only the push-type tasks need it."""
    n = len(obs0['observation'])
    if task in ('push',):
        return obs0['observation'][:, 3] <= -0.45
    if task == 'chest_push':
        return obs0['observation'][:, 8] <= -0.45
    if task == 'block_rearrange':
        return (obs0['observation'][:, 8] <= -0.45) & (obs0['observation'][:, 24] <= -0.45)
    return np.ones(n, bool)


def run(name, backend, N, seed=0):
    task = name.rsplit('_', 1)[0] if name.startswith(('block_stack_', 'block_rearrange_')) else name
    kw, T = SUITE[name]
    if backend == 'device':
        import pybullet_multigoal_gym_amd as pmg
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            env = pmg.make_env(task=task, num_envs=N, seed=seed, seed_stride=1, max_episode_steps=T, **kw)
        obs = env.reset()
    else:
        import oracle_lib
        env = oracle_lib.OracleEnv(task, N, seed_base=seed, seed_stride=1, threads=oracle_lib.usable_threads(), max_episode_steps=T,
                                   f32=(backend == 'oracle_f32'), **kw)
        env.reset()
        obs = env.reset()
    pol = SP.make_policy(task, N, **({'num_block': kw['num_block']} if 'num_block' in kw else {}))
    feas = feasible_mask(task, obs)
    lists = {'prone': 0, 'free': 0, 'redo': 0}

    def on_step(t, a, o, r, ok):
        if backend == 'device':
            for k, v in env.handle.schedule().items():
                lists[k] += len(v)
    t0 = time.time()
    obs, ok, ever = SP.rollout(env, pol, T, obs, on_step)
    out = {'task': name, 'backend': backend, 'N': N, 'T': T, 'success_end': float(ok.mean()), 'success_ever': float(ever.mean()),
           'feasible': int(feas.sum()), 'success_ever_feasible': float(ever[feas].mean()) if feas.any() else None,
           'seconds': round(time.time() - t0, 2)}
    if backend == 'device':
        out['env_steps_one_env_lists'] = lists['prone']
        out['env_steps_fast_paths'] = lists['free']
        out['env_steps_redo'] = lists['redo']
    env.close()
    return out, ever


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'oracle'
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    names = sys.argv[3].split(',') if len(sys.argv) > 3 else list(SUITE)
    for name in names:
        for backend in (['device', 'oracle'] if which == 'both' else [which]):
            print(json.dumps(run(name, backend, N)[0]), flush=True)
