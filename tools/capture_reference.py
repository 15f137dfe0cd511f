#!/usr/bin/env python3
"""Capture golden rollouts FROM THE REFERENCE ITSELF.

Cannot run in the build container or on the GPU box: it needs pybullet~=3.0.6 and gym~=0.17.3, which are absent
in both (that absence is why the oracle's header says "parity unpinned").  Run it wherever those two packages and
a checkout of IanYangChina/pybullet_multigoal_gym exist:

    PYTHONPATH=/path/to/pybullet_multigoal_gym python tools/capture_reference.py --out tests/fixtures

It imports the reference there only, and writes small .npz fixtures (inputs + outputs, no reference source):
per task, seed 0, the action table RandomState(12345).uniform(-1, 1, [T, A]).astype(float32), and per step every
observation array, reward, goal_achieved, the joint state, the tip target and the block base poses.
tests/test_reference_fixtures.py replays them through the oracle (and, with -m gpu, the HIP library) and pins
parity to the 1e-3 of BASELINE.json; without fixtures that test is skipped.
"""
import argparse
import os

import numpy as np

CONFIGS = [('reach', {}), ('push', {}), ('pick_and_place', {}), ('pick_and_place', {'binary_reward': False}),
           ('slide', {}), ('block_stack', {'num_block': 4}), ('block_rearrange', {'num_block': 3}),
           ('chest_push', {'num_block': 2}), ('chest_pick_and_place', {'num_block': 2})]


def capture(task, kw, T):
    import pybullet_multigoal_gym as pmg          # the REFERENCE package
    env = pmg.make_env(task=task, gripper='parallel_jaw', render=False, max_episode_steps=T, **kw)
    env.seed(0)
    obs = env.reset()
    A = env.action_space.shape[0]
    actions = np.random.RandomState(12345).uniform(-1, 1, (T, A)).astype(np.float32)
    inner = env.unwrapped
    rec = {k: [np.asarray(obs[k], np.float64)] for k in ('observation', 'policy_state', 'achieved_goal', 'desired_goal')}
    rec.update(reward=[], goal_achieved=[], done=[], joint_state=[], tip_target=[], block_poses=[])

    def extras():
        rec['joint_state'].append(np.asarray(inner.robot.get_kuka_joint_state()[0], np.float64))
        rec['tip_target'].append(np.asarray(inner.robot.end_effector_target, np.float64))
        poses = []
        for name, body in sorted(inner.object_bodies.items()):
            if body is not None and name.startswith('block'):
                p, q = inner._p.getBasePositionAndOrientation(body)
                v, w = inner._p.getBaseVelocity(body)
                poses.append(np.concatenate([p, q, v, w]))
        rec['block_poses'].append(np.asarray(poses, np.float64).reshape(-1))
    extras()
    for t in range(T):
        obs, r, d, info = env.step(actions[t])
        for k in ('observation', 'policy_state', 'achieved_goal', 'desired_goal'):
            rec[k].append(np.asarray(obs[k], np.float64))
        rec['reward'].append(float(r)); rec['goal_achieved'].append(bool(info['goal_achieved'])); rec['done'].append(bool(d))
        extras()
    env.close()
    out = {k: np.asarray(v) for k, v in rec.items()}
    out['actions'] = actions
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'fixtures'))
    ap.add_argument('--steps', type=int, default=50)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for task, kw in CONFIGS:
        tag = task + ''.join('_%s%s' % (k[:3], int(v) if not isinstance(v, bool) else int(v)) for k, v in sorted(kw.items()))
        data = capture(task, kw, args.steps)
        np.savez_compressed(os.path.join(args.out, 'ref_%s.npz' % tag), task=task, kwargs=repr(kw), **data)
        print('wrote', tag, {k: v.shape for k, v in data.items()})


if __name__ == '__main__':
    main()
