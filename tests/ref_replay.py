"""Replay of tests/golden/ref_*.json -- sessions recorded from the REFERENCE's own Python (tools/gen_reference_fixtures.py)
-- through the oracle's env entry points or through the product (HIP library / CPU emulator of its kernels).

A fixture is a list of events (reset / step / seed / set_sub_goal / sub_goals / curriculum_update / compute_reward) with
everything the reference returned.  `replay()` issues the same calls and compares every output; what may differ is the
float tolerance: the oracle shares the fixture's physics (float64), so it must agree to float32 output rounding; the
device computes in float32 with a different formulation, so its trajectories get the bars of DESIGN.md section 5.
"""
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ('observation', 'policy_state', 'achieved_goal', 'desired_goal')


def fixture_paths():
    return sorted(glob.glob(os.path.join(HERE, 'golden', 'ref_*.json')))


def load(path):
    with open(path) as f:
        return json.load(f)


class OracleAdapter:
    """oracle/libpmg_oracle.so behind the call shapes of the reference env (one env, reference seeding)."""

    def __init__(self, fx, f32=False):
        import oracle_lib
        self.env = oracle_lib.OracleEnv(fx['task'], 1, f32=f32, seed_base=0, seed_stride=0, **fx['oracle_kwargs'])
        self.env.reset()                       # the constructor's reset (base_env.py:84)
        self.nb = fx['oracle_kwargs'].get('num_block', 4)
        self.fx = fx

    def reset(self):
        return {k: v[0] for k, v in self.env.reset().items()}

    def step(self, a):
        o, r, d, ok = self.env.step(np.asarray(a, np.float32)[None])
        return {k: v[0] for k, v in o.items()}, r[0], bool(d[0]), bool(ok[0])

    def seed(self, s):
        self.env.seed(s, 0)

    def set_sub_goal(self, k):
        self.env.set_sub_goal(k)
        # desired goal of the current state without stepping: a masked reset of nobody returns everybody's observation
        return self.env.reset(mask=np.zeros(1, np.uint8))['desired_goal'][0]

    def curriculum_update(self, on):
        self.env.curriculum_update(on)

    def curriculum(self):
        c = self.env.curriculum()
        return dict(level=int(c['level'][0]), goal_step=int(c['goal_step'][0]), prob=c['prob'][0], generated=c['generated'][0])

    def compute_reward(self, ag, dg):
        return self.env.compute_reward(ag, dg)

    def state(self):
        return self.env.get_state()[0]

    def sub_goals(self):
        return None

    def close(self):
        self.env.close()


class ProductAdapter:
    """pybullet_multigoal_gym_amd.make_env(...) un-batched (num_envs=None: the reference's shapes), any library."""

    def __init__(self, fx, library=None):
        import warnings
        import pybullet_multigoal_gym_amd as pmg
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            self.env = pmg.make_env(task=fx['task'], gripper='parallel_jaw', render=False, num_envs=None, seed=0, seed_stride=0,
                                    _library=library, **fx['make_kwargs'])
        self.fx = fx

    def reset(self):
        return self.env.reset()

    def step(self, a):
        o, r, d, info = self.env.step(np.asarray(a, np.float32))
        return o, r, d, info['goal_achieved']

    def seed(self, s):
        self.env.seed(s)

    def set_sub_goal(self, k):
        return self.env.set_sub_goal(k)

    def curriculum_update(self, on):
        (self.env.activate_curriculum_update if on else self.env.deactivate_curriculum_update)()

    def curriculum(self):
        e = self.env
        return dict(level=int(e.last_curriculum_level), goal_step=int(e.curriculum_goal_step), prob=e.curriculum_prob,
                    generated=e.num_generated_goals_per_curriculum)

    def compute_reward(self, ag, dg):
        return self.env._compute_reward(np.asarray(ag, np.float32), np.asarray(dg, np.float32))

    def state(self):
        return self.env.get_state()[0]

    def sub_goals(self):
        return self.env.sub_goals

    def close(self):
        self.env.close()


def _cmp(tag, got, want, tol):
    got, want = np.asarray(got, np.float64).reshape(-1), np.asarray(want, np.float64).reshape(-1)
    assert got.shape == want.shape, '%s: shape %s != %s' % (tag, got.shape, want.shape)
    err = float(np.abs(got - want).max()) if got.size else 0.0
    assert err <= tol, '%s: max |got - reference| = %.3g > %.3g\n got  %s\n want %s' % (tag, err, tol, got, want)
    return err


def _state_fields(state, nb, chest):
    st = dict(q=state[0:9], qd=state[9:18], ee_target=state[18:21], joint_target=state[21:28],
              blocks=[state[64 + 13 * b:77 + 13 * b] for b in range(nb)])
    if chest:
        st['door'] = state[48:50]
    return st


def replay(fx, env, tol_static=2e-6, tol_traj=2e-6, tol_vel=None, traj_steps=None, threshold_guard=0.0, check_internal=True):
    """Issue the fixture's calls on `env` and compare every output with what the reference returned.

    tol_static: outputs that involve no time stepping (reset observations, goals, sub-goals, curricula);
    tol_traj / tol_vel: positions / velocities after stepSimulation (velocity entries of the observation are those beyond
    the position block); traj_steps: compare trajectories only for the first n steps after each reset (contact-rich
    float32 rollouts decorrelate); threshold_guard: skip reward / success flags when the distance is this close to the
    threshold.  Returns the largest deviations seen."""
    task = fx['task']
    nb = 0 if task == 'reach' else (fx['oracle_kwargs'].get('num_block', 5) if task.startswith(('block', 'chest')) else 1)
    chest = task.startswith('chest')
    thr = fx['make_kwargs'].get('distance_threshold', 0.05)
    binary = fx['make_kwargs'].get('binary_reward', True)
    tol_vel = tol_traj if tol_vel is None else tol_vel
    worst = dict(static=0.0, traj=0.0, obs=0.0)
    since_reset = 0
    died = False
    for i, ev in enumerate(fx['events']):
        op, out = ev['op'], ev.get('out')
        tag = '%s event %d (%s)' % (task, i, op)
        if op == 'reset':
            if 'error' in out:        # the reference raised (curriculum budget exhausted out of order): the session ends
                died = True
                break
            o = env.reset()
            since_reset = 0
            for k in KEYS:
                if k in out['obs']:      # sampling sessions record the goals only
                    worst['static'] = max(worst['static'], _cmp(tag + ' ' + k, o[k], out['obs'][k], tol_static))
            if out.get('curriculum') is not None:
                c = env.curriculum()
                want = out['curriculum']
                if 'level' in want:
                    assert c['level'] == want['level'], (tag, c['level'], want['level'])
                assert c['goal_step'] == want['goal_step'], (tag, c['goal_step'], want['goal_step'])
                _cmp(tag + ' curriculum_prob', c['prob'], want['prob'], 0)
                _cmp(tag + ' generated', c['generated'], want['generated'], 0)
            if check_internal:
                st = _state_fields(env.state(), nb, chest)
                want = out['internal']
                for k in ('q', 'qd', 'ee_target', 'joint_target'):
                    if k in want:
                        _cmp(tag + ' ' + k, st[k], want[k], tol_static)
                for b in range(nb):
                    _cmp(tag + ' block %d' % b, st['blocks'][b], want['blocks'][b], tol_static)
                if 'order' in want and task == 'block_stack':   # the stacking order of this episode (state row 40..44)
                    got = [int(v) for v in env.state()[40:40 + nb]]
                    assert got == want['order'], (tag, got, want['order'])
        elif op == 'step':
            o, r, d, ok = env.step(ev['action'])
            since_reset += 1
            assert bool(d) == out['done'], (tag, d, out['done'])
            check_traj = traj_steps is None or since_reset <= traj_steps
            if check_traj:
                for k in KEYS:      # 'observation' holds velocities too: its own bar and its own worst
                    t = tol_vel if k == 'observation' else tol_traj
                    w = 'obs' if k == 'observation' else 'traj'
                    worst[w] = max(worst[w], _cmp(tag + ' ' + k, o[k], out['obs'][k], t))
                dist = float(np.linalg.norm(np.asarray(out['obs']['achieved_goal']) - np.asarray(out['obs']['desired_goal'])))
                if abs(dist - thr) > threshold_guard:
                    assert bool(ok) == out['goal_achieved'], (tag, ok, out['goal_achieved'], dist)
                    if binary:
                        rr, want = np.float32(r), np.float32(out['reward'])
                        assert rr == want and np.signbit(rr) == np.signbit(want), (tag, r, out['reward'])   # incl. float32 -0.0
                    else:
                        _cmp(tag + ' reward', r, out['reward'], max(tol_traj, 1e-6))
                if check_internal:
                    st = _state_fields(env.state(), nb, chest)
                    want = out['internal']
                    for k in ('q', 'ee_target', 'joint_target'):
                        _cmp(tag + ' ' + k, st[k], want[k], tol_traj)
                    _cmp(tag + ' qd', st['qd'], want['qd'], tol_vel)
                    if chest:
                        _cmp(tag + ' door', st['door'][:1], want['door'][:1], tol_traj)
        elif op == 'seed':
            env.seed(ev['seed'])
        elif op == 'set_sub_goal':
            if 'error' in out:
                continue              # the reference fails its own shape assert here (chest sub-goal 0 without grip goals)
            g = env.set_sub_goal(ev['ind'])
            want = np.asarray(out['goal'])[:len(np.asarray(g).reshape(-1))]   # chest sub-goal 0 is over-long in the reference
            worst['static'] = max(worst['static'], _cmp(tag, g, want, tol_static if since_reset == 0 else tol_traj))
        elif op == 'sub_goals':
            gs = env.sub_goals()
            if gs is None:
                continue
            want = out['goals']
            assert len(gs) == len(want), (tag, len(gs), len(want))
            for k, (g, w) in enumerate(zip(gs, want)):
                w = np.asarray(w)[:len(np.asarray(g).reshape(-1))]   # chest sub-goal 0 is over-long in the reference
                _cmp(tag + ' [%d]' % k, g, w, tol_static if since_reset == 0 else tol_traj)
        elif op == 'curriculum_update':
            env.curriculum_update(ev['enabled'])
        elif op == 'compute_reward':
            ag, dg = np.asarray(ev['ag']), np.asarray(ev['dg'])
            r, ok = env.compute_reward(ag, dg)
            d64 = np.linalg.norm(ag - dg, axis=-1)
            d32 = np.linalg.norm(ag.astype(np.float32).astype(np.float64) - dg.astype(np.float32).astype(np.float64), axis=-1)
            sure = (np.abs(d64 - thr) > 1e-6) & ((d64 > thr) == (d32 > thr))     # float32 transport of the goals may flip a tie
            assert np.array_equal(np.asarray(ok, bool)[sure], np.asarray(out['goal_achieved'], bool)[sure]), tag
            want = np.asarray(out['reward'], np.float64)
            if binary:
                got = np.asarray(r, np.float32)
                assert np.array_equal(got[sure], want.astype(np.float32)[sure]) and np.array_equal(np.signbit(got[sure]), np.signbit(want[sure])), tag
            else:
                _cmp(tag, np.asarray(r)[sure], want[sure], 1e-6)
        else:
            raise ValueError(op)
    worst['died'] = died
    return worst
