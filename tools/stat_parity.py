#!/usr/bin/env python3
"""Statistical parity at scale (GPU box): device vs float64 oracle against the oracle's own float32-vs-float64 spread over a
long random-policy episode -- percentiles of the block-position error per step, and how many blocks left the table."""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib  # noqa: E402
import pybullet_multigoal_gym_amd as pmg  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else 'block_stack'
N, T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024, int(sys.argv[3]) if len(sys.argv) > 3 else 50
kw = {'num_block': 4} if task not in ('push', 'slide', 'pick_and_place', 'reach') else {}
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    env = pmg.make_env(task=task, num_envs=N, seed=0, seed_stride=1, **kw)
o64 = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=16, **kw)
o32 = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=16, f32=True, **kw)
o64.reset(), o32.reset()
env.reset(), o64.reset(), o32.reset()
rs = np.random.RandomState(12345)
A = env.dims.action_dim
for t in range(T):
    a = rs.uniform(-1, 1, (N, A)).astype(np.float32)
    o = env.step(a)[0]
    a64 = o64.step(a)[0]
    a32 = o32.step(a)[0]
    if (t + 1) % 10 == 0 or t == T - 1:
        err = np.abs(o['achieved_goal'] - a64['achieved_goal']).max(1)
        spr = np.abs(a32['achieved_goal'] - a64['achieved_goal']).max(1)
        c0 = 1 if task.startswith('chest') else 0
        z = lambda x: x['achieved_goal'][:, c0:c0 + 3 * kw.get('num_block', 1)].reshape(N, -1, 3)[..., 2]
        print(json.dumps({'task': task, 'step': t + 1, 'dev_vs_f64_p50_p90_p99': [float(np.percentile(err, q)) for q in (50, 90, 99)],
                          'f32_vs_f64_p50_p90_p99': [float(np.percentile(spr, q)) for q in (50, 90, 99)],
                          'fell_dev_f64_f32': [int((z(x) < 0.1).sum()) for x in (o, a64, a32)]}))
