"""Stand-in modules for the reference's third-party imports (see the package docstring).

gym 0.17.3 is restated from its published source ([GYM-PRIOR]): seeding.np_random = RandomState seeded with the
32-bit words of sha512(str(seed))[:8]; registry.make = entry point(**kwargs) wrapped in TimeLimit when the spec has
max_episode_steps; TimeLimit.step sets done and info['TimeLimit.truncated'] once elapsed >= max; Wrapper forwards
public attributes; Box.contains = shape and bounds.
"""
import copy
import hashlib
import importlib
import struct
import sys
import types

import numpy as np

REFERENCE_ROOT = '/root/reference'


# ----------------------------------------------------------------------------- gym.utils.seeding
def _bigint_from_bytes(b):
    sizeof_int = 4
    padding = sizeof_int - len(b) % sizeof_int
    b += b'\0' * padding
    int_count = int(len(b) / sizeof_int)
    unpacked = struct.unpack('{}I'.format(int_count), b)
    accum = 0
    for i, val in enumerate(unpacked):
        accum += 2 ** (sizeof_int * 8 * i) * val
    return accum


def _int_list_from_bigint(bigint):
    if bigint < 0:
        raise ValueError('Seed must be non-negative, not {}'.format(bigint))
    elif bigint == 0:
        return [0]
    ints = []
    while bigint > 0:
        bigint, mod = divmod(bigint, 2 ** 32)
        ints.append(mod)
    return ints


def hash_seed(seed=None, max_bytes=8):
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    h = hashlib.sha512(str(seed).encode('utf8')).digest()
    return _bigint_from_bytes(h[:max_bytes])


def create_seed(a=None, max_bytes=8):
    if a is None:
        import os
        a = _bigint_from_bytes(os.urandom(max_bytes))
    elif isinstance(a, str):
        a = a.encode('utf8')
        a += hashlib.sha512(a).digest()
        a = _bigint_from_bytes(a[:max_bytes])
    elif isinstance(a, int):
        a = a % 2 ** (8 * max_bytes)
    else:
        raise ValueError('Invalid type for seed: {} ({})'.format(type(a), a))
    return a


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, int) and 0 <= seed):
        raise ValueError('Seed must be a non-negative integer or omitted, not {}'.format(seed))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(_int_list_from_bigint(hash_seed(seed)))
    return rng, seed


# ----------------------------------------------------------------------------- gym core / spaces / registration
class Env(object):
    metadata = {'render.modes': []}
    reward_range = (-float('inf'), float('inf'))
    spec = None
    action_space = None
    observation_space = None

    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        return

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = self.env.action_space
        self.observation_space = self.env.observation_space
        self.reward_range = self.env.reward_range
        self.metadata = self.env.metadata

    def __getattr__(self, name):
        if name.startswith('_'):
            raise AttributeError("attempted to get missing private attribute '{}'".format(name))
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def seed(self, seed=None):
        return self.env.seed(seed)

    def close(self):
        return self.env.close()

    def compute_reward(self, achieved_goal, desired_goal, info):
        return self.env.compute_reward(achieved_goal, desired_goal, info)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps=None):
        super(TimeLimit, self).__init__(env)
        if max_episode_steps is None and self.env.spec is not None:
            max_episode_steps = env.spec.max_episode_steps
        if self.env.spec is not None:
            self.env.spec.max_episode_steps = max_episode_steps
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        assert self._elapsed_steps is not None, 'Cannot call env.step() before calling reset()'
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info['TimeLimit.truncated'] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            assert low.shape == high.shape, 'box dimension mismatch'
            self.shape = low.shape
            self.low, self.high = low, high
        else:
            assert np.isscalar(low) and np.isscalar(high)
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low)
            self.high = np.full(self.shape, high)
        self.low = self.low.astype(self.dtype)
        self.high = self.high.astype(self.dtype)

    def contains(self, x):
        if isinstance(x, list):
            x = np.array(x)
        return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)


class Dict(Space):
    def __init__(self, spaces=None, **kw):
        self.spaces = dict(spaces or kw)
        self.shape = None
        self.dtype = None


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)


class EnvSpec(object):
    def __init__(self, id, entry_point=None, max_episode_steps=None, kwargs=None):
        self.id = id
        self.entry_point = entry_point
        self.max_episode_steps = max_episode_steps
        self._kwargs = {} if kwargs is None else kwargs
        self.tags = {}

    def make(self, **kwargs):
        _kwargs = self._kwargs.copy()
        _kwargs.update(kwargs)
        mod_name, attr_name = self.entry_point.split(':')
        cls = getattr(importlib.import_module(mod_name), attr_name)
        env = cls(**_kwargs)
        spec = copy.deepcopy(self)
        spec._kwargs = _kwargs
        env.unwrapped.spec = spec
        return env


class EnvRegistry(object):
    def __init__(self):
        self.env_specs = {}

    def register(self, id, **kwargs):
        if id in self.env_specs:
            raise RuntimeError('Cannot re-register id: {}'.format(id))
        self.env_specs[id] = EnvSpec(id, **kwargs)

    def make(self, id, **kwargs):
        spec = self.env_specs[id]
        env = spec.make(**kwargs)
        if env.spec.max_episode_steps is not None:
            env = TimeLimit(env, max_episode_steps=env.spec.max_episode_steps)
        return env


registry = EnvRegistry()


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Put the stand-ins into sys.modules and the reference on sys.path.  Refuses to shadow a real gym / pybullet."""
    for real in ('gym', 'pybullet'):
        if real in sys.modules and not getattr(sys.modules[real], '__pmg_stub__', False):
            raise RuntimeError('%s is really installed here: capture fixtures with `tools/gen_reference_fixtures.py --real` instead' % real)
    from . import fake_bullet
    seeding = _module('gym.utils.seeding', np_random=np_random, hash_seed=hash_seed, create_seed=create_seed)
    utils = _module('gym.utils', seeding=seeding)
    spaces = _module('gym.spaces', Box=Box, Dict=Dict, MultiDiscrete=MultiDiscrete, Space=Space)
    registration = _module('gym.envs.registration', register=lambda id, **kw: registry.register(id, **kw),
                           make=lambda id, **kw: registry.make(id, **kw), registry=registry, EnvSpec=EnvSpec)
    envs = _module('gym.envs', registration=registration)
    wrappers = _module('gym.wrappers', TimeLimit=TimeLimit)
    _module('gym', Env=Env, Wrapper=Wrapper, utils=utils, spaces=spaces, envs=envs, wrappers=wrappers,
            make=registration.make, __pmg_stub__=True)

    class error(Exception):
        pass
    pb = _module('pybullet', error=error, __pmg_stub__=True, **fake_bullet.CONSTANTS)
    bc = _module('pybullet_utils.bullet_client', BulletClient=fake_bullet.FakeBulletClient)
    _module('pybullet_utils', bullet_client=bc)

    def _unsupported(*a, **k):
        raise NotImplementedError('numpy-quaternion is only used by end-effector rotation control (outside the hot path)')
    _module('quaternion', as_float_array=_unsupported, from_euler_angles=_unsupported)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return pb
