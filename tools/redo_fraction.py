#!/usr/bin/env python3
"""Diagnostics: how many envs per step leave the packed path (redo list) over a random-policy episode."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybullet_multigoal_gym_amd as pmg
task = sys.argv[1] if len(sys.argv) > 1 else 'push'
N = 4096
env = pmg.make_env(task=task, num_envs=N, seed=0, seed_stride=1, num_block=4)
rs = np.random.RandomState(12345)
env.reset()
fr = []
pr = []
for t in range(50):
    env.step(rs.uniform(-1, 1, (N, env.dims.action_dim)).astype(np.float32))
    sc = env.handle.schedule()
    fr.append(len(sc['redo'])); pr.append(len(sc['prone']))
print(task, 'redo per step:', fr[::5], 'mean', np.mean(fr), 'max', max(fr), '| one-env-per-wave list:', pr[::5], 'max', max(pr))
