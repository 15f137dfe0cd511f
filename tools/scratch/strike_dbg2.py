import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import oracle_lib
speed = float(sys.argv[1])
env = oracle_lib.OracleEnv('slide', 4, seed_base=0, seed_stride=1, threads=4, max_episode_steps=400)
env.reset(); obs = env.reset()
st = env.get_state().copy()
# puck in front of the tip along -x, tip low
st[:, 64] = st[:, 18] - 0.06; st[:, 65] = st[:, 19]; st[:, 66] = 0.17; st[:, 67:71] = [0,0,0,1]; st[:, 71:77] = 0
env.set_state(st)
for t in range(40):
    a = np.zeros((4, 3), np.float32); a[:, 0] = -speed
    obs = env.step(a)[0]
    ob = obs['observation'].astype(np.float64)
    pv = ob[:, 10:13] - ob[:, 14:17]
    if abs(pv[0,0]) > 1e-4 or t % 10 == 0: print(t, 'tip x %.4f puck x %.4f gap %.4f puck vx %.4f tip vx %.4f' % (ob[0,0], ob[0,3], ob[0,0]-ob[0,3], pv[0,0], ob[0,10]))
