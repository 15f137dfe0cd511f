import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import oracle_lib, scripted_policies as SP
T=130
env = oracle_lib.OracleEnv('slide', 8, seed_base=0, seed_stride=1, threads=8, max_episode_steps=T)
env.reset(); obs = env.reset()
phase = np.zeros(8, np.int32)
saved = None
for t in range(83):
    ob = obs['observation'].astype(np.float64)
    tip, puck = ob[:, 0:3], ob[:, 3:6]
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
    if t == 82:
        saved = env.get_state().copy(); act = a.copy()
        break
    obs = env.step(a)[0]
e = 1
s = saved
print('state env1: q', s[e, :9].round(4), 'tip target', s[e, 18:21], 'puck pos', s[e, 64:67], 'quat', s[e, 67:71], 'vel', s[e, 71:77])
print('tip y %.4f puck y %.4f' % (ob[e, 1], ob[e, 4]), 'action', act[e])
for scale in (1.0, 0.5, 0.25, 0.1, 0.0):
    env.set_state(saved)
    a = act.copy(); a[:, 0] *= scale
    o = env.step(a)[0]['observation']
    print('a_x %.2f -> puck x %.4f (was %.4f), tip x %.4f' % (-scale, o[e, 3], s[e, 64], o[e, 0]))
