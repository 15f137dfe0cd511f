"""Minimal stand-ins for gym.spaces.Box / Dict (gym is not a dependency).

Mirrors what the reference exposes through gym 0.17.3 (P/robots/kuka.py:103-118,
P/envs/base_envs/base_env.py:85-110): shape, dtype, bounds, contains(), sample().
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            low = np.asarray(low)
            shape = low.shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, self.dtype), self.shape).copy()
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi, self.shape).astype(self.dtype)

    def __repr__(self):
        return 'Box(%s, %s)' % (self.shape, self.dtype)


class Dict:
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def __repr__(self):
        return 'Dict(%s)' % ', '.join('%s:%r' % kv for kv in self.spaces.items())
