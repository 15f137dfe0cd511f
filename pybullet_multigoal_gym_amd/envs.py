"""Batched Kuka multigoal environments behind the reference's env interface.

Host-side mirror of the reference's gym.Env surface for the hot path:
``reset() / step() / seed() / close() / _compute_reward() / action_space /
observation_space`` (P/envs/base_envs/base_env.py:120-138, P/__init__.py:4-178),
with a leading ``num_envs`` axis.  All arithmetic happens in the HIP library
behind the C ABI (include/pmg.h); this file only validates arguments, moves
numpy buffers and reproduces gym's TimeLimit bookkeeping.

Documented deviations from the reference (DESIGN.md "API deviations"):
  * arrays are float32 (the reference returns float64); pass ``dtype=np.float64``
    to up-cast at the boundary;
  * ``num_envs=None`` gives the reference's un-batched shapes, ``num_envs=N``
    adds a leading axis to every array, reward, done and info entry;
  * env *i* is seeded with ``seed + i*seed_stride`` (``seed_stride=0``
    reproduces the reference, where every env is seeded 0);
  * no auto-reset: like the reference, the caller resets (optionally per env
    with ``reset(mask=...)``).
"""
import warnings

import numpy as np

from . import spaces
from ._lib import PmgHandle, TASK_IDS, default_library

TASK_CLASS_NAMES = {'reach': 'KukaReachEnv', 'push': 'KukaPushEnv', 'pick_and_place': 'KukaPickAndPlaceEnv',
                    'slide': 'KukaSlideEnv', 'block_stack': 'KukaBlockStackEnv',
                    'block_rearrange': 'KukaBlockRearrangeEnv', 'chest_push': 'KukaChestPushEnv',
                    'chest_pick_and_place': 'KukaChestPickAndPlaceEnv'}


class KukaVecEnv:
    """N Kuka iiwa14 + parallel-jaw worlds on one MI355X (one wavefront per env)."""

    metadata = {'render.modes': [], 'video.frames_per_second': 25}

    def __init__(self, task='reach', num_envs=None, binary_reward=True, joint_control=False, max_episode_steps=50,
                 distance_threshold=0.05, num_block=5, random_order=True, seed=0, seed_stride=1, device=0,
                 env_index_offset=0, dtype=np.float32, task_decomposition=False, use_curriculum=False,
                 num_goals_to_generate=1e6, grip_informed_goal=False, _library=None):
        if task not in TASK_IDS:
            raise ValueError('invalid task name: {}, only support: {}'.format(task, sorted(TASK_IDS)))
        self.task = task
        self.batched = num_envs is not None
        self.num_envs = int(num_envs) if self.batched else 1
        self.binary_reward = bool(binary_reward)
        self.joint_control = bool(joint_control)
        self.distance_threshold = float(distance_threshold)
        self._max_episode_steps = int(max_episode_steps)
        self.num_block = int(num_block)
        self.task_decomposition = bool(task_decomposition)
        self.curriculum = bool(use_curriculum)
        chest = task in ('chest_push', 'chest_pick_and_place')
        multi = task in ('block_stack', 'block_rearrange') or chest
        if self.task_decomposition:   # kuka_multi_step_base_env.py:123, kuka_multi_step_envs.py:159
            assert not self.curriculum, 'if using task decomposition, curriculum should be False, vice versa'
            assert task == 'block_stack' or chest, 'task decomposition is accelerated for block_stack and the chest tasks'
        if self.curriculum:
            assert multi, 'curriculum is a multi-block task option'
            assert self.num_block >= 2 or chest, 'the curriculum schedule needs at least two levels'
            warnings.warn("You will need to call env.activate_curriculum_update() before your training phase, "
                          "and env.deactivate_curriculum_update() before your evaluation phase.")
        self.curriculum_update = False
        self.grip_informed_goal = bool(grip_informed_goal)
        if self.grip_informed_goal:   # kuka_multi_step_envs.py:13-17,158
            assert task == 'block_stack' or chest, 'gripper informed goals are accelerated for block_stack and the chest tasks'
        self.num_steps = self.num_block * (2 if self.grip_informed_goal else 1) if multi else None
        if chest:                     # kuka_multi_step_envs.py:238-242, 388-392
            per_block = (3 if task == 'chest_pick_and_place' else 2) if self.grip_informed_goal else 1
            self.num_steps = self.num_block * per_block + 1
        self.dtype = np.dtype(dtype)
        self._seed_stride = int(seed_stride)
        self.handle = PmgHandle(_library or default_library(), task=TASK_IDS[task], num_envs=self.num_envs,
                                num_block=self.num_block, binary_reward=int(self.binary_reward),
                                joint_control=int(self.joint_control), max_episode_steps=self._max_episode_steps,
                                device=int(device), distance_threshold=self.distance_threshold,
                                random_order=int(bool(random_order)), seed_base=int(seed), seed_stride=int(seed_stride),
                                env_index_offset=int(env_index_offset), task_decomposition=int(self.task_decomposition),
                                use_curriculum=int(self.curriculum), num_goals_to_generate=int(num_goals_to_generate),
                                grip_informed_goal=int(self.grip_informed_goal))
        d = self.handle.dims
        self.dims = d
        self.action_space = spaces.Box(-np.ones([d.action_dim]), np.ones([d.action_dim]))
        # the reference consumes one reset in its constructor (base_env.py:84) to size the spaces
        obs = self._wrap_obs(*self.handle.reset())
        box = lambda k: spaces.Box(-np.inf, np.inf, shape=obs[k].shape[-1:], dtype='float32')
        self.observation_space = spaces.Dict(dict(
            observation=box('observation'), state=box('observation'),  # 'state' is the reference's key (base_env.py:86-92)
            policy_state=box('policy_state'), achieved_goal=box('achieved_goal'), desired_goal=box('desired_goal')))
        self._needs_reset = True   # gym TimeLimit: step() before reset() is an error
        self._closed = False

    # ------------------------------------------------------------------
    @property
    def dt(self):
        """As the reference's property: timestep * frame_skip = 0.04 s (base_env.py:112-118).
        One env.step() simulates 5 of those (kuka.py:223-225): see ``sim_time_per_step``."""
        return 0.002 * 20

    sim_time_per_step = 0.2

    def _wrap_obs(self, o, p, a, g):
        out = {'observation': o, 'policy_state': p, 'achieved_goal': a, 'desired_goal': g}
        if self.dtype != np.float32:
            out = {k: v.astype(self.dtype) for k, v in out.items()}
        if not self.batched:
            out = {k: v[0] for k, v in out.items()}
        return out

    def seed(self, seed=None):
        """base_env.py:120-122 (gym.utils.seeding.np_random): env i gets seed + i*seed_stride."""
        if seed is None:
            seed = int(np.random.SeedSequence().generate_state(1)[0])
        if not (isinstance(seed, (int, np.integer)) and seed >= 0):
            raise ValueError('Seed must be a non-negative integer or omitted, not {}'.format(seed))
        self.handle.seed(int(seed), self._seed_stride)
        return [int(seed)]

    def reset(self, test=False, mask=None):
        """base_env.py:124-128.  ``mask`` ([N] bool) resets a subset and returns everybody's observation."""
        if mask is not None and not self.batched:
            raise ValueError('mask is only meaningful with num_envs=N')
        obs = self._wrap_obs(*self.handle.reset(mask))
        self._needs_reset = False
        return obs

    def step(self, action):
        """TimeLimit.step + base_env.py:130-138 -> (obs, reward, done, info)."""
        assert not self._needs_reset, 'Cannot call env.step() before calling reset()'
        d = self.dims
        a = np.asarray(action, dtype=np.float32)
        want = (self.num_envs, d.action_dim) if self.batched else (d.action_dim,)
        assert a.shape == want, 'action shape {} != {}'.format(a.shape, want)
        a = np.ascontiguousarray(a.reshape(self.num_envs, d.action_dim))
        assert np.all(a >= -1.0) and np.all(a <= 1.0), 'action outside action_space Box(-1, 1)'  # kuka.py:168
        o, p, ag, dg, r, ok, dn = self.handle.step(a)
        obs = self._wrap_obs(o, p, ag, dg)
        if not self.binary_reward and self.dtype != np.float32:
            r = r.astype(self.dtype)
        if self.batched:
            return obs, r, dn, {'goal_achieved': ok, 'TimeLimit.truncated': dn.copy()}
        info = {'goal_achieved': bool(ok[0])}
        if dn[0]:
            info['TimeLimit.truncated'] = True   # inner env never sets done (base_env.py:138)
        return obs, r[0], bool(dn[0]), info

    def _compute_reward(self, achieved_goal, desired_goal):
        """kuka_single_step_base_env.py:237-244 on arrays of shape [..., goal_dim] (HER relabelling)."""
        ag = np.asarray(achieved_goal)
        dg = np.asarray(desired_goal)
        assert ag.shape == dg.shape
        r, ok = self.handle.compute_reward(ag, dg)
        if ag.ndim == 1:
            return r.reshape(()), ok.reshape(())
        return r, ok

    compute_reward = _compute_reward

    # reference multi-step API (kuka_multi_step_base_env.py:142-177); one curriculum / sub-goal state PER ENV,
    # as N separate reference envs would have
    def activate_curriculum_update(self):
        if not self.curriculum:
            warnings.warn('This method should not be called while not using curriculum.')
            return
        self.curriculum_update = True
        self.handle.curriculum_update(True)

    def deactivate_curriculum_update(self):
        if not self.curriculum:
            warnings.warn('This method should not be called while not using curriculum.')
            return
        self.curriculum_update = False
        self.handle.curriculum_update(False)

    def set_sub_goal(self, sub_goal_ind, mask=None):
        """kuka_multi_step_base_env.py:154-177: switch the desired goal to sub-goal ``sub_goal_ind`` (-1 = the
        final goal) for all / the masked envs; returns the desired goals."""
        if not self.task_decomposition:
            warnings.warn('The set_sub_goal() method should only be called when using task decomposition,\n'
                          'It does nothing and returns None when self.task_decomposition is False.')
            return None
        self.handle.set_sub_goal(sub_goal_ind, mask)
        g = self.handle.read_outputs()[3]
        g = g.astype(self.dtype) if self.dtype != np.float32 else g
        return g if self.batched else g[0]

    @property
    def sub_goals(self):
        """The reference's ``self.sub_goals`` list (kuka_multi_step_envs.py:89-122): entry k keeps the first k+1
        blocks of the stacking order at their targets and every other block where it currently is."""
        if not self.task_decomposition:
            return None
        st = self.handle.get_state()
        if self.task in ('chest_push', 'chest_pick_and_place'):
            # kuka_multi_step_envs.py:285-342, 433-475: the goals depend on the live block / gripper poses; read each
            # one off the library and put the active indices (state column 39) back
            levels = st[:, 39].astype(int)
            out = []
            for k in range(self.num_steps):
                self.handle.set_sub_goal(k, None)
                g = self.handle.read_outputs()[3]
                g = g.astype(self.dtype) if self.dtype != np.float32 else g.copy()
                out.append(g if self.batched else g[0])
            for lv in np.unique(levels):
                self.handle.set_sub_goal(int(lv), levels == lv)
            return out
        nb, n = self.num_block, len(st)
        rows = np.arange(n)
        order = st[:, 40:40 + nb].astype(int)
        targets = st[:, 48:48 + 3 * nb].reshape(-1, nb, 3)
        blocks = np.stack([st[:, 64 + 13 * b:67 + 13 * b] for b in range(nb)], axis=1)

        def goal(n_at_target, grip=None):
            g = blocks.copy()
            for i in range(n_at_target):
                g[rows, order[:, i]] = targets[rows, order[:, i]]
            g = g.reshape(n, 3 * nb)
            if grip is not None:
                g = np.concatenate([g, grip, np.full((n, 1), 0.03, np.float32)], axis=1)
            g = g.astype(self.dtype)
            return g if self.batched else g[0]

        out = []
        for k in range(nb):
            if self.grip_informed_goal:   # (pick, place) per block: kuka_multi_step_envs.py:91-111
                out.append(goal(k, blocks[rows, order[:, k]]))
                out.append(goal(k + 1, targets[rows, order[:, k]]))
            else:
                out.append(goal(k + 1))
        return out

    def _curriculum(self, idx):
        if not self.curriculum:
            return None
        v = self.handle.curriculum_read()[idx]
        return v if self.batched else v[0]

    last_curriculum_level = property(lambda self: self._curriculum(0))
    curriculum_goal_step = property(lambda self: self._curriculum(1))
    curriculum_prob = property(lambda self: self._curriculum(2))
    num_generated_goals_per_curriculum = property(lambda self: self._curriculum(3))

    def render(self, mode='human', camera_id=0):
        raise NotImplementedError('rendering / image observations are outside the accelerated hot path (state obs only)')

    # checkpoint / test hooks
    def get_state(self):
        return self.handle.get_state()

    def set_state(self, state):
        self.handle.set_state(state)
        self._needs_reset = False

    def get_checkpoint(self):
        """Everything a later set_checkpoint() needs to continue bit-identically: the state rows, the per-env MT19937
        streams (future goals / orders / curriculum draws) and the host-side curriculum switch."""
        return {'state': self.handle.get_state(), 'rng': self.handle.get_rng(), 'curriculum_update': self.curriculum_update}

    def set_checkpoint(self, ck):
        self.handle.set_state(ck['state'])
        self.handle.set_rng(ck['rng'])
        if self.curriculum:
            self.curriculum_update = bool(ck['curriculum_update'])
            self.handle.curriculum_update(self.curriculum_update)
        self._needs_reset = False

    def set_goal(self, goals, mask=None):
        self.handle.set_goal(goals, mask)

    def close(self):
        if not self._closed:
            self.handle.close()
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
