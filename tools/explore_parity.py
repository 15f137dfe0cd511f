"""Exploration helper (GPU box): HIP vs oracle(f64) error statistics next to the oracle's own f32-vs-f64 spread."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import pybullet_multigoal_gym_amd as pmg
import oracle_lib as O

def run(task, N=64, T=50, **kw):
    env = pmg.make_env(task=task, num_envs=N, seed=0, seed_stride=1, **kw)
    okw = {k: v for k, v in kw.items() if k in ('num_block', 'joint_control', 'binary_reward')}
    o64 = O.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=16, **okw); o64.reset()
    o32 = O.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=16, f32=True, **okw); o32.reset()
    env.reset(); o64.reset(); o32.reset()
    rs = np.random.RandomState(12345); A = env.dims.action_dim
    t0 = time.time()
    for t in range(T):
        a = rs.uniform(-1, 1, (N, A)).astype(np.float32)
        o, r, d, info = env.step(a); oo, ro, do, ok = o64.step(a); o3, r3, d3, ok3 = o32.step(a)
        if t in (0, 4, 9, 24, 49):
            e = np.abs(o['observation'] - oo['observation']).max(1); e3 = np.abs(o3['observation'] - oo['observation']).max(1)
            print('%-14s %s t=%2d  HIP-vs-f64 obs: med %.2e p90 %.2e max %.2e | f32-vs-f64 oracle: med %.2e p90 %.2e max %.2e | rew mism %d' % (
                task, kw, t, np.median(e), np.percentile(e, 90), e.max(), np.median(e3), np.percentile(e3, 90), e3.max(), int((r != ro).sum())), flush=True)
    env.close()

if __name__ == '__main__':
    run('reach'); run('reach', joint_control=True); run('push'); run('pick_and_place'); run('block_stack', num_block=4)
