#!/usr/bin/env python3
"""Can the slide puck be STRUCK to its goal?  (kuka_single_step_envs.py:49-59: goals lie 0.05-0.6 m beyond the -x edge of the
tip's clip box, kuka_single_step_base_env.py:53-56,66-69.)  A scripted striker: rise, fly to the run-up point behind the puck
(+x side, on the puck's y), descend, then a = (-1, hold y, hold z) EVERY step -- the tip accelerates through the puck until
its target hits the clip box's edge at x = -0.67 -- and wait for the puck to come to rest.  Reports, per env, the puck's
speed when the tip stops, the distance it coasts after that, its final position against the goal; as distributions.
    tools/strike_puck.py [N] [run_up_m] [--device]      (CPU oracle by default; --device: the HIP library on a GPU box)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import scripted_policies as SP  # noqa: E402


def run(N, run_up, device, T=130):
    if device:
        import pybullet_multigoal_gym_amd as pmg
        env = pmg.make_env(task='slide', num_envs=N, seed=0, seed_stride=1, max_episode_steps=T)
    else:
        import oracle_lib
        env = oracle_lib.OracleEnv('slide', N, seed_base=0, seed_stride=1, threads=oracle_lib.usable_threads(), max_episode_steps=T)
        env.reset()
    obs = env.reset()
    phase = np.zeros(N, np.int32)
    x_stop = np.full(N, np.nan)      # puck x / speed at the step the tip stops advancing
    v_stop = np.full(N, np.nan)
    vmax = np.zeros(N)
    ever = np.zeros(N, bool)
    still = np.zeros(N, np.int32)
    for t in range(T):
        ob = obs['observation'].astype(np.float64)
        # 20 columns: tip 0:3, puck 3:6, finger closeness 6, tip - puck 7:10, tip velocity 10:13, finger velocity 13, tip - puck
        # velocity 14:17, angular 17:20 (kuka_single_step_base_env.py:193-211)
        tip, puck, pv = ob[:, 0:3], ob[:, 3:6], ob[:, 10:13] - ob[:, 14:17]
        a = np.zeros((N, 3), np.float32)
        start = np.stack([np.minimum(puck[:, 0] + run_up, SP.TIP_HIGH[0] - 0.004), puck[:, 1], np.full(N, 0.222)], 1)
        m = phase == 0                                               # rise / fly to the run-up point
        tgt = start.copy()
        low = tip[:, 2] < 0.222 - 0.012
        near = np.abs(tip[:, :2] - puck[:, :2]).max(1) < 0.06
        tgt[low & near, :2] = tip[low & near, :2]
        a[m] = np.clip((tgt - tip)[m] / SP.STEP, -1, 1)
        phase[m & (np.abs(start[:, :2] - tip[:, :2]).max(1) < 0.005)] = 1
        m = phase == 1                                               # descend
        tgt = start.copy(); tgt[:, 2] = 0.177
        a[m] = np.clip((tgt - tip)[m] / SP.STEP, -1, 1)
        phase[m & (tip[:, 2] < 0.181)] = 2
        m = phase == 2                                               # strike
        a[m, 0] = -1.0
        a[m, 1] = np.clip((puck[:, 1] - tip[:, 1])[m] / SP.STEP, -1, 1) * 0.0     # straight line: no lateral correction during the run
        a[m, 2] = np.clip((0.177 - tip[:, 2])[m] / SP.STEP, -1, 1)
        at_edge = m & (tip[:, 0] < SP.TIP_LOW[0] + 0.004)
        x_stop[at_edge & np.isnan(x_stop)] = puck[at_edge & np.isnan(x_stop), 0]
        v_stop[at_edge & np.isnan(v_stop)] = -pv[at_edge & np.isnan(v_stop), 0]
        phase[at_edge] = 3
        vmax = np.maximum(vmax, np.where(phase >= 2, -pv[:, 0], 0.0))
        out = env.step(a)
        obs = out[0]
        ok = out[3]['goal_achieved'] if isinstance(out[3], dict) else out[3]
        ever |= np.asarray(ok, bool)
    ob = obs['observation'].astype(np.float64)
    puck = ob[:, 3:6]
    goal = obs['desired_goal'].astype(np.float64)
    struck = ~np.isnan(x_stop)
    coast = np.where(struck, x_stop - puck[:, 0], np.nan)
    beyond = SP.TIP_LOW[0] - puck[:, 0]            # how far past the tip box's edge the puck ends
    need = SP.TIP_LOW[0] - goal[:, 0]              # how far past the edge its goal lies
    pc = lambda v, q: [float(np.nanpercentile(v, x)) for x in q]
    res = {'who': 'device' if device else 'oracle', 'N': N, 'T': T, 'run_up_m': run_up, 'envs_that_struck': int(struck.sum()),
           'puck_speed_when_the_tip_stops_p50_p90_max': pc(v_stop, (50, 90, 100)), 'puck_peak_speed_p50_p90_max': pc(vmax, (50, 90, 100)),
           'coast_after_the_tip_stops_m_p50_p90_max': pc(coast, (50, 90, 100)),
           'puck_end_beyond_the_tip_box_edge_m_p50_p90_max': pc(beyond, (50, 90, 100)),
           'goal_beyond_the_tip_box_edge_m_p10_p50_p90': pc(need, (10, 50, 90)),
           'goals_within_the_reached_band': float((need <= np.nanmax(beyond) + 0.05).mean()),
           'phase_histogram_at_the_end': np.bincount(phase, minlength=4).tolist(), 'success_ever': float(ever.mean()), 'success_at_the_end': float(np.asarray(ok, bool).mean())}
    env.close()
    return res


if __name__ == '__main__':
    args = [x for x in sys.argv[1:] if not x.startswith('--')]
    N = int(args[0]) if args else 256
    ups = [float(args[1])] if len(args) > 1 else [0.10, 0.20, 0.28]
    for ru in ups:
        print(json.dumps(run(N, ru, '--device' in sys.argv)), flush=True)
