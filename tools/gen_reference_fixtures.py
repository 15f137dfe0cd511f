#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE's own code: tests/golden/ref_*.json.

Build container only (reads /root/reference; nothing of it is copied -- the JSON holds inputs and outputs).  For every
accelerated task and option the reference's real `pybullet_multigoal_gym.make_env(...)` builds its real env objects on
top of tools/refharness (stand-in gym, scripted Bullet client backed by the oracle's physics), a scripted session runs
(reset / step / seed / set_sub_goal / curriculum switches / _compute_reward), and every output is recorded:

    python tools/gen_reference_fixtures.py            # rewrites tests/golden/ref_*.json

What this pins: env ids and spaces, the constructor's resets, seeding
and draw order, object / goal sampling, stack orders, curricula and their probability schedules, sub-goal lists, the
float32 action product and clipping, motor commands and their order, the 5 x stepSimulation cadence, observation
assembly for every layout, rewards (incl. float32 -0.0), goal_achieved, TimeLimit.  What it cannot pin: the physics
under stepSimulation / calculateInverseKinematics (row a21), which is the oracle's own [BULLET-PRIOR] restatement on
both sides of the comparison.

numpy note: the reference pins numpy~=1.19, where float32_scalar + python_float promotes to float64 (value-based
casting); numpy 2.2 here would keep float32 (NEP 50) in kuka.py:171 `(a[-1] + 1.0) * ...`.  Actions are therefore
handed to the reference as an ndarray subclass whose scalar indexing yields float64 -- slices stay float32, so
kuka.py:205/209's float32 array products are untouched -- which reproduces the pinned numpy's arithmetic.
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from refharness import fake_bullet, stubs  # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')
KEYS = ('observation', 'policy_state', 'achieved_goal', 'desired_goal')


class LegacyF32(np.ndarray):
    """float32 action vector with numpy-1.19 scalar promotion (see the module docstring)."""

    def __getitem__(self, i):
        v = np.ndarray.__getitem__(self, i)
        if isinstance(v, np.ndarray):
            return v.view(np.ndarray)
        return np.float64(v)


def flt(x):
    return [float(v) for v in np.asarray(x, np.float64).reshape(-1)]


def obs_rec(o):
    return {k: flt(o[k]) for k in KEYS}


def curriculum_rec(inner):
    if not getattr(inner, 'curriculum', False):
        return None
    r = dict(goal_step=int(inner.curriculum_goal_step), prob=flt(inner.curriculum_prob),
             generated=flt(inner.num_generated_goals_per_curriculum))
    if getattr(inner, 'last_curriculum_level', None) is not None:
        r['level'] = int(inner.last_curriculum_level)
    if getattr(inner, 'last_ind_block_to_move', None) is not None:
        r['moved'] = [int(i) for i in inner.last_ind_block_to_move]
    return r


REAL = False   # --real: the reference on REAL pybullet / gym (wherever those exist); files real_*.json


def real_world_state(inner):
    """What FakeBulletClient.world_state() reads from the oracle, read from a real Bullet client instead."""
    p, robot = inner._p, inner.robot
    idx = list(robot.kuka_joint_index) + list(robot.gripper_joint_index)
    js = [p.getJointState(robot.kuka_body_index, j) for j in idx]
    st = dict(q=[float(s_[0]) for s_ in js], qd=[float(s_[1]) for s_ in js], blocks=[])
    names = ['block'] if 'block' in inner.object_bodies else [k for k in getattr(inner, 'block_keys', [])][:getattr(inner, 'num_block', 0)]
    for k in names:
        body = inner.object_bodies.get(k)
        if body is None:
            continue
        pos, orn = p.getBasePositionAndOrientation(body)
        lin, ang = p.getBaseVelocity(body)
        st['blocks'].append([float(v) for v in tuple(pos) + tuple(orn) + tuple(lin) + tuple(ang)])
    if getattr(inner, 'chest', False):
        dq, dqd, _ = inner.chest_robot.jdict[inner.chest_robot.door_joint_name].get_state()
        st['door'] = [float(dq), float(dqd)]
    return st


def internal_rec(inner):
    st = real_world_state(inner) if REAL else fake_bullet.LAST_CLIENT.world_state()
    st['ee_target'] = flt(inner.robot.end_effector_target)
    st['joint_target'] = flt(inner.robot.joint_state_target)
    if getattr(inner, 'last_order', None) is not None:
        st['order'] = [int(i) for i in inner.last_order]
    return st


def run_session(task, make_kw, oracle_kw, script, light=False):
    import pybullet_multigoal_gym as ref
    if not REAL:
        fake_bullet.configure(task=task, **oracle_kw)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        env = ref.make_env(task=task, gripper='parallel_jaw', render=False, **make_kw)
    inner = env.unwrapped
    client = None if REAL else fake_bullet.LAST_CLIENT
    A = env.action_space.shape[0]
    fx = dict(task=task, make_kwargs=make_kw, oracle_kwargs=oracle_kw, env_id=inner.spec.id,
              max_episode_steps=env._max_episode_steps, action_dim=A,
              action_low=flt(env.action_space.low), action_high=flt(env.action_space.high),
              observation_space={k: list(v.shape) for k, v in env.observation_space.spaces.items()},
              world_params={} if REAL else {k: (list(v) if isinstance(v, tuple) else v) for k, v in client.params.items()},
              after_constructor=dict(internal=internal_rec(inner), curriculum=curriculum_rec(inner)), events=[])
    ev = fx['events']
    rs = np.random.RandomState(12345)
    for op in script:
        kind = op[0]
        if kind == 'reset':
            try:
                o = env.reset()
            except ValueError as ex:
                # numpy's "probabilities do not sum to 1": a later curriculum level used up its budget before an earlier one
                # (possible with tiny num_goals_to_generate).  The reference dies here; so does the recorded session.
                ev.append(dict(op='reset', out=dict(error=str(ex))))
                break
            if light:   # sampling sessions: what a reset DRAWS (goal, object poses, order, curriculum), not the whole observation
                st = internal_rec(inner)
                ev.append(dict(op='reset', out=dict(obs={k: flt(o[k]) for k in ('achieved_goal', 'desired_goal')},
                                                    internal={k: st[k] for k in ('blocks', 'order') if k in st}, curriculum=curriculum_rec(inner))))
            else:
                ev.append(dict(op='reset', out=dict(obs=obs_rec(o), internal=internal_rec(inner), curriculum=curriculum_rec(inner))))
        elif kind == 'step':
            n = op[1]
            bias = np.asarray(op[2], np.float32) if len(op) > 2 else np.zeros(A, np.float32)
            for _ in range(n):
                a = np.clip(rs.uniform(-1, 1, A).astype(np.float32) + bias, -1, 1).astype(np.float32)
                o, r, d, info = env.step(a.view(LegacyF32))
                assert isinstance(r, (np.floating, float)), type(r)
                ev.append(dict(op='step', action=flt(a), out=dict(
                    obs=obs_rec(o), reward=float(r), reward_dtype=str(np.asarray(r).dtype), done=bool(d),
                    goal_achieved=bool(info['goal_achieved']), truncated=bool(info.get('TimeLimit.truncated', False)),
                    internal=internal_rec(inner))))
        elif kind == 'seed':
            ret = env.seed(op[1])
            ev.append(dict(op='seed', seed=op[1], out=[int(s) for s in ret]))
        elif kind == 'set_sub_goal':
            try:
                g = env.set_sub_goal(op[1])
                ev.append(dict(op='set_sub_goal', ind=op[1], out=dict(goal=flt(g))))
            except Exception as ex:   # noqa: BLE001 -- a reference failure is a recorded behaviour
                ev.append(dict(op='set_sub_goal', ind=op[1], out=dict(error=type(ex).__name__)))
        elif kind == 'sub_goals':
            ev.append(dict(op='sub_goals', out=dict(goals=[flt(g) for g in inner.sub_goals], sub_goal_ind=int(inner.sub_goal_ind))))
        elif kind == 'curriculum_update':
            (env.activate_curriculum_update if op[1] else env.deactivate_curriculum_update)()
            ev.append(dict(op='curriculum_update', enabled=bool(op[1])))
        elif kind == 'compute_reward':
            B = op[1]
            G = inner.desired_goal.shape[0]
            rr = np.random.RandomState(777)
            ag = rr.uniform(-0.7, 0.3, (B, G))
            dg = ag + rr.uniform(-1, 1, (B, G)) * rr.choice([0.0, 0.01, 0.03, 0.1], size=(B, 1))
            r, ok = inner._compute_reward(ag, dg)
            ev.append(dict(op='compute_reward', ag=[flt(x) for x in ag], dg=[flt(x) for x in dg],
                           out=dict(reward=flt(r), reward_dtype=str(r.dtype), goal_achieved=[bool(x) for x in ok])))
        else:
            raise ValueError(kind)
    fx['bullet_calls'] = {} if REAL else dict(client.calls)
    fx['urdf_fk_max_err'] = 0.0 if REAL else client.fk_max_err
    fx['physics'] = 'pybullet (the real reference)' if REAL else 'oracle/pmg_oracle.c behind a scripted Bullet client'
    env.close()
    return fx


DOWN3 = [0, 0, -0.6]
SESSIONS = [
    # name, task, make_env kwargs, oracle config kwargs, script
    ('reach', 'reach', {}, {}, [('reset',), ('step', 50), ('reset',), ('step', 4), ('seed', 123), ('reset',), ('step', 3), ('compute_reward', 24)]),
    ('reach_dense', 'reach', dict(binary_reward=False), dict(binary_reward=False), [('reset',), ('step', 8)]),
    ('reach_joint', 'reach', dict(joint_control=True), dict(joint_control=True), [('reset',), ('step', 12)]),
    ('reach_short', 'reach', dict(max_episode_steps=5, distance_threshold=0.1), dict(max_episode_steps=5, distance_threshold=0.1),
     [('reset',), ('step', 6), ('reset',), ('step', 2)]),
    ('push', 'push', {}, {}, [('reset',), ('step', 50), ('reset',), ('step', 4), ('seed', 7), ('reset',), ('step', 2)]),
    ('push_joint', 'push', dict(joint_control=True), dict(joint_control=True), [('reset',), ('step', 8)]),
    ('pick_and_place', 'pick_and_place', {}, {}, [('reset',), ('step', 30, [0, 0, -0.5, 0]), ('reset',), ('step', 3)]),
    ('pick_and_place_dense', 'pick_and_place', dict(binary_reward=False), dict(binary_reward=False), [('reset',), ('step', 10), ('compute_reward', 16)]),
    ('slide', 'slide', {}, {}, [('reset',), ('step', 30), ('reset',), ('step', 3)]),
    ('block_stack4', 'block_stack', dict(num_block=4), dict(num_block=4), [('reset',), ('step', 30, [0, 0, -0.5, 0]), ('reset',), ('step', 3), ('compute_reward', 16)]),
    ('block_stack5_grip', 'block_stack', dict(num_block=5, grip_informed_goal=True), dict(num_block=5, grip_informed_goal=True),
     [('reset',), ('step', 10, [0, 0, -0.5, 0]), ('reset',), ('step', 2)]),
    ('block_stack3_decomp', 'block_stack', dict(num_block=3, task_decomposition=True), dict(num_block=3, task_decomposition=True),
     [('reset',), ('sub_goals',), ('set_sub_goal', 0), ('step', 3), ('set_sub_goal', 1), ('step', 3), ('sub_goals',), ('set_sub_goal', 2), ('step', 2),
      ('set_sub_goal', -1), ('step', 2), ('reset',), ('step', 2)]),
    ('block_stack2_decomp_grip', 'block_stack', dict(num_block=2, task_decomposition=True, grip_informed_goal=True),
     dict(num_block=2, task_decomposition=True, grip_informed_goal=True),
     [('reset',), ('sub_goals',), ('set_sub_goal', 0), ('step', 2), ('set_sub_goal', 1), ('step', 2), ('set_sub_goal', 2), ('step', 2), ('set_sub_goal', 3), ('step', 2), ('sub_goals',)]),
    ('block_stack4_curriculum', 'block_stack', dict(num_block=4, use_curriculum=True, num_goals_to_generate=24),
     dict(num_block=4, use_curriculum=True, num_goals_to_generate=24),
     [('reset',), ('step', 2), ('curriculum_update', True)] + [('reset',)] * 26 + [('step', 2), ('curriculum_update', False), ('reset',), ('reset',)]),
    ('block_stack3_curriculum_grip', 'block_stack', dict(num_block=3, use_curriculum=True, grip_informed_goal=True, num_goals_to_generate=12),
     dict(num_block=3, use_curriculum=True, grip_informed_goal=True, num_goals_to_generate=12),
     [('curriculum_update', True)] + [('reset',), ('step', 1)] * 10),
    ('block_rearrange3', 'block_rearrange', dict(num_block=3), dict(num_block=3), [('reset',), ('step', 20), ('reset',), ('step', 2)]),
    ('block_rearrange4_curriculum', 'block_rearrange', dict(num_block=4, use_curriculum=True, num_goals_to_generate=24),
     dict(num_block=4, use_curriculum=True, num_goals_to_generate=24),
     [('curriculum_update', True)] + [('reset',)] * 26 + [('step', 3)]),
    ('chest_push2', 'chest_push', dict(num_block=2), dict(num_block=2), [('reset',), ('step', 30), ('reset',), ('step', 3)]),
    ('chest_push3_grip_decomp', 'chest_push', dict(num_block=3, grip_informed_goal=True, task_decomposition=True),
     dict(num_block=3, grip_informed_goal=True, task_decomposition=True),
     [('reset',), ('sub_goals',)] + sum([[('set_sub_goal', k), ('step', 1)] for k in range(7)], []) + [('set_sub_goal', -1), ('step', 2), ('sub_goals',)]),
    ('chest_push2_decomp', 'chest_push', dict(num_block=2, task_decomposition=True), dict(num_block=2, task_decomposition=True),
     [('reset',), ('set_sub_goal', 1), ('step', 2), ('set_sub_goal', 2), ('step', 2), ('set_sub_goal', 0), ('set_sub_goal', -1), ('step', 1)]),
    ('chest_push3_curriculum_grip', 'chest_push', dict(num_block=3, use_curriculum=True, grip_informed_goal=True, num_goals_to_generate=16),
     dict(num_block=3, use_curriculum=True, grip_informed_goal=True, num_goals_to_generate=16),
     [('reset',), ('step', 2), ('curriculum_update', True)] + [('reset',), ('step', 1)] * 18),
    ('chest_pick_and_place2', 'chest_pick_and_place', dict(num_block=2), dict(num_block=2), [('reset',), ('step', 30, [0, 0, -0.4, 0]), ('reset',), ('step', 3)]),
    ('chest_pick_and_place2_grip_decomp', 'chest_pick_and_place', dict(num_block=2, grip_informed_goal=True, task_decomposition=True),
     dict(num_block=2, grip_informed_goal=True, task_decomposition=True),
     [('reset',), ('sub_goals',)] + sum([[('set_sub_goal', k), ('step', 1)] for k in range(7)], []) + [('set_sub_goal', -1), ('step', 2)]),
    ('chest_pick_and_place5_curriculum', 'chest_pick_and_place', dict(num_block=5, use_curriculum=True, num_goals_to_generate=18),
     dict(num_block=5, use_curriculum=True, num_goals_to_generate=18),
     [('curriculum_update', True)] + [('reset',)] * 20 + [('step', 2)]),
]


def sampling(seeds, resets, pre=()):
    return list(pre) + sum([[('seed', sd)] + [('reset',)] * resets for sd in seeds], [])


SEEDS = [0, 1, 2, 3, 12345, 2 ** 31 + 7]
CUR = dict(use_curriculum=True, num_goals_to_generate=40)
# many-seed sampling sessions (light records): every draw of every task's reset comes from the reference's own
# _task_reset / _generate_goal / _generate_curriculum -- nothing about the sampling rules is re-typed anywhere in this repo
SAMPLING = [
    ('sampling_reach', 'reach', {}, {}, sampling(SEEDS, 5)),
    ('sampling_push', 'push', {}, {}, sampling(SEEDS, 5)),
    ('sampling_pick_and_place', 'pick_and_place', {}, {}, sampling(SEEDS, 8)),
    ('sampling_slide', 'slide', {}, {}, sampling(SEEDS, 5)),
    ('sampling_block_stack5', 'block_stack', dict(num_block=5), dict(num_block=5), sampling(SEEDS, 5)),
    ('sampling_block_stack2', 'block_stack', dict(num_block=2), dict(num_block=2), sampling(SEEDS[:3], 4)),
    ('sampling_block_stack4_curriculum', 'block_stack', dict(num_block=4, **CUR), dict(num_block=4, **CUR),
     sampling(SEEDS[:4], 12, pre=[('curriculum_update', True)])),
    ('sampling_block_rearrange5', 'block_rearrange', dict(num_block=5), dict(num_block=5), sampling(SEEDS, 4)),
    ('sampling_block_rearrange3_curriculum', 'block_rearrange', dict(num_block=3, **CUR), dict(num_block=3, **CUR),
     sampling(SEEDS[:4], 12, pre=[('curriculum_update', True)])),
    ('sampling_chest_push4', 'chest_push', dict(num_block=4), dict(num_block=4), sampling(SEEDS, 4)),
    ('sampling_chest_push2_curriculum', 'chest_push', dict(num_block=2, **CUR), dict(num_block=2, **CUR),
     sampling(SEEDS[:4], 12, pre=[('curriculum_update', True)])),
    ('sampling_chest_pick_and_place3_curriculum_grip', 'chest_pick_and_place', dict(num_block=3, grip_informed_goal=True, **CUR),
     dict(num_block=3, grip_informed_goal=True, **CUR), sampling(SEEDS[:4], 12, pre=[('curriculum_update', True)])),
]


def main():
    global REAL
    REAL = '--real' in sys.argv
    if REAL:
        # wherever pybullet~=3.0.6 and gym~=0.17.3 exist (not this container, not the GPU box): the same sessions on the
        # REAL reference -> tests/golden/real_*.json, which tests/test_reference_golden.py then holds the oracle and the
        # device to at BASELINE.json's 1e-3.  UNTESTED here for want of those packages.
        import gym, pybullet  # noqa: F401, E401
        sys.path.insert(0, os.environ.get('PMG_REFERENCE_ROOT', '/root/reference'))
    else:
        stubs.install()
    only = set(a for a in sys.argv[1:] if not a.startswith('--'))
    os.makedirs(OUT, exist_ok=True)
    total = 0
    for name, task, mk, ok, script in SESSIONS + SAMPLING:
        if only and name not in only:
            continue
        import gym
        gym.envs.registration.registry.env_specs.clear()   # the reference registers by id: same id, different kwargs otherwise
        if REAL and name.startswith('sampling_'):
            continue
        fx = run_session(task, mk, ok, script, light=name.startswith('sampling_'))
        path = os.path.join(OUT, ('real_%s.json' if REAL else 'ref_%s.json') % name)
        with open(path, 'w') as f:
            json.dump(fx, f, separators=(',', ':'))
        total += os.path.getsize(path)
        print('%-36s %4d events %7d bytes  fk_err %.1e  %s' % (name, len(fx['events']), os.path.getsize(path), fx['urdf_fk_max_err'], fx['env_id']))
    print('total %d bytes' % total)


if __name__ == '__main__':
    main()
