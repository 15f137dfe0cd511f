#!/usr/bin/env python3
"""Device-resident throughput of a task under joint control (not a BASELINE configuration; diagnostics)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybullet_multigoal_gym_amd as pmg
task = sys.argv[1] if len(sys.argv) > 1 else 'reach'
N = 4096
env = pmg.make_env(task=task, num_envs=N, joint_control=True, seed=0, seed_stride=1)
env.reset()
A = env.dims.action_dim
acts = np.random.RandomState(1).uniform(-1, 1, (60, N, A)).astype(np.float32)
h = env.handle
d = h.device_alloc(acts.nbytes)
h.upload(d, acts)
for t in range(10):
    h.step_device(d + t * N * A * 4)
h.sync()
t0 = time.perf_counter()
for t in range(10, 60):
    if t == 35:
        h.reset_device(None)
    h.step_device(d + t * N * A * 4)
h.sync()
el = time.perf_counter() - t0
sc = h.schedule()
print(task, 'joint_control PMG_PACKED=%s' % os.environ.get('PMG_PACKED', '1'), round(N * 50 / el), 'env-steps/s | lists:',
      len(sc['prone']), 'one-env,', len(sc['free']), 'fast,', len(sc['redo']), 'redo')
