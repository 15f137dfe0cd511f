import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import oracle_lib
N = 64
env = oracle_lib.OracleEnv('slide', N, seed_base=0, seed_stride=1, threads=8, max_episode_steps=400)
env.reset(); obs = env.reset()
st = env.get_state().copy()
runup = float(sys.argv[1]) if len(sys.argv) > 1 else 0.10
st[:, 64] = st[:, 18] - 0.0445 - runup; st[:, 65] = st[:, 19]; st[:, 66] = 0.17; st[:, 67:71] = [0,0,0,1]; st[:, 71:77] = 0
env.set_state(st)
speeds = np.linspace(0.1, 1.0, N)
x0 = st[:, 64].copy()
tipx_end = None
for t in range(150):
    ob = obs['observation'] if t else None
    a = np.zeros((N, 3), np.float32)
    tipx = env.get_state()[:, 18]
    a[:, 0] = np.where(tipx > -0.665, -speeds, 0.0)
    obs = env.step(a)[0]
px = obs['observation'][:, 3]
for i in range(0, N, 4):
    print('a %.2f  puck travelled %.3f m, ends %.3f beyond the edge' % (speeds[i], x0[i] - px[i], -0.67 - px[i]))
