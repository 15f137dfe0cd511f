#!/usr/bin/env python3
"""Long random-policy soak on the GPU: every task, 2048 envs, 300 steps with resets every 50; reports non-finite
values, objects leaving the workspace, and the success / contact statistics of the rollout."""
import sys, os, json, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybullet_multigoal_gym_amd as pmg

N, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2048, 300)
for task, kw in [('reach', {}), ('reach', {'joint_control': True}), ('push', {}), ('slide', {}), ('pick_and_place', {}),
                 ('block_stack', {'num_block': 4}), ('block_rearrange', {'num_block': 4}), ('push', {'joint_control': True}),
                 ('chest_push', {'num_block': 4}), ('chest_pick_and_place', {'num_block': 4})]:
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        env = pmg.make_env(task=task, num_envs=N, seed=1, seed_stride=1, **kw)
    rs = np.random.RandomState(0)
    A = env.dims.action_dim
    bad = 0; succ = 0; zmin = 9.0; zmax = -9.0; far = 0
    for t in range(T):
        if t % 50 == 0:
            env.reset()
        o, r, d, info = env.step(rs.uniform(-1, 1, (N, A)).astype(np.float32))
        bad += int((~np.isfinite(o['observation'])).sum())
        succ += int(info['goal_achieved'].sum())
        if task != 'reach':
            ag = o['achieved_goal'][:, (1 if task.startswith('chest') else 0):].reshape(N, -1, 3)
            zmin = min(zmin, float(ag[..., 2].min())); zmax = max(zmax, float(ag[..., 2].max()))
            far += int((np.abs(ag[..., 0] + 0.6) > 0.6).sum() + (np.abs(ag[..., 1]) > 0.5).sum())
    print(json.dumps({'task': task, **kw, 'nonfinite': bad, 'success_rate': succ / (N * T), 'obj_z_range': [zmin, zmax], 'left_workspace': far}))
    env.close()
