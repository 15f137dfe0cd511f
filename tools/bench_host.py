#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer path: env.step(numpy actions) -> numpy observations.
Never the bench `value`; it tells a numpy-only RL loop what it gets."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybullet_multigoal_gym_amd as pmg

task = sys.argv[1] if len(sys.argv) > 1 else 'reach'
N, K = 4096, 100
env = pmg.make_env(task=task, num_envs=N, num_block=4)
A = env.dims.action_dim
acts = np.random.RandomState(0).uniform(-1, 1, (K + 10, N, A)).astype(np.float32)
env.reset()
for t in range(10):
    env.step(acts[t])
t0 = time.perf_counter()
for t in range(10, 10 + K):
    if (t - 10) % 50 == 0:
        env.reset()
    env.step(acts[t])
el = time.perf_counter() - t0
print(json.dumps({'task': task, 'path': 'host buffers (H2D actions, kernel, D2H packed rows, unpack)', 'env_steps_per_s': N * K / el,
                  'ms_per_step': el / K * 1e3, 'bytes_h2d_per_step': N * A * 4, 'bytes_d2h_per_step': N * env.dims.packed_dim * 4}))
