import os, sys, numpy as np
sys.path.insert(0, '.')
import pybullet_multigoal_gym_amd as pmg
task = sys.argv[1]; kw = eval(sys.argv[2]) if len(sys.argv) > 2 else {}
which = sys.argv[3] if len(sys.argv) > 3 else '0'
def make(p):
    os.environ['PMG_PACKED'] = p
    try:
        return pmg.make_env(task=task, num_envs=512, seed=5, seed_stride=1, **kw)
    finally:
        del os.environ['PMG_PACKED']
envs = [make(p) for p in which]
for e in envs: e.reset()
rs = np.random.RandomState(9)
for t in range(12):
    a = rs.uniform(-1, 1, (512, envs[0].dims.action_dim)).astype(np.float32)
    for p, e in zip(which, envs):
        e.step(a)
        sch = e.handle.schedule()
        print(task, 'packed', p, 'step', t, 'ok', {k: len(v) for k, v in sch.items()} if p == '1' else '', flush=True)
