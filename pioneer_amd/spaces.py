"""Minimal stand-in for ``gym.spaces.Box`` (gym is not a dependency of the engine).

Only what PioneerKinematicEnv's callers use: low/high/shape/dtype, sample(),
contains().  If gym is importable, ``to_gym()`` converts to the real class.
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            low = np.asarray(low)
            high = np.asarray(high)
            shape = low.shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1e3)
        hi = np.where(np.isfinite(self.high), self.high, 1e3)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    __contains__ = contains

    def to_gym(self):
        from gym import spaces  # noqa: only if the caller has gym
        return spaces.Box(self.low, self.high, dtype=self.dtype)

    def __repr__(self):
        return f"Box({self.shape}, {self.dtype})"
