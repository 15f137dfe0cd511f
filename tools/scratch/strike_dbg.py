import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import oracle_lib, scripted_policies as SP
N=1; T=130
env = oracle_lib.OracleEnv('slide', 8, seed_base=0, seed_stride=1, threads=8, max_episode_steps=T)
env.reset(); obs = env.reset()
phase = np.zeros(8, np.int32)
for t in range(T):
    ob = obs['observation'].astype(np.float64)
    tip, puck, pv = ob[:, 0:3], ob[:, 3:6], ob[:, 10:13] - ob[:, 14:17]
    a = np.zeros((8, 3), np.float32)
    start = np.stack([np.minimum(puck[:, 0] + 0.2, SP.TIP_HIGH[0] - 0.004), puck[:, 1], np.full(8, 0.222)], 1)
    m = phase == 0
    tgt = start.copy(); low = tip[:, 2] < 0.21; near = np.abs(tip[:, :2] - puck[:, :2]).max(1) < 0.06
    tgt[low & near, :2] = tip[low & near, :2]
    a[m] = np.clip((tgt - tip)[m] / SP.STEP, -1, 1)
    phase[m & (np.abs(start[:, :2] - tip[:, :2]).max(1) < 0.005)] = 1
    m = phase == 1
    tgt = start.copy(); tgt[:, 2] = 0.177
    a[m] = np.clip((tgt - tip)[m] / SP.STEP, -1, 1)
    phase[m & (tip[:, 2] < 0.181)] = 2
    m = phase == 2
    a[m, 0] = -1.0; a[m, 2] = np.clip((0.177 - tip[:, 2])[m] / SP.STEP, -1, 1)
    phase[m & (tip[:, 0] < SP.TIP_LOW[0] + 0.004)] = 3
    e = 1
    print(t, 'ph', phase[e], 'tip x %.4f z %.4f vx %.3f' % (tip[e,0], tip[e,2], ob[e,10]), 'puck x %.4f z %.4f vx %.3f vz %.3f' % (puck[e,0], puck[e,2], pv[e,0], pv[e,2]), 'gap %.4f' % (tip[e,0]-puck[e,0]))
    obs = env.step(a)[0]
