"""Minimal Box space so the env surface works without gym installed (the reference uses
gym.spaces.Box, src/gym/network_sim.py:377-388).  If gym or gymnasium is importable their
Box is returned instead, so agents that type-check the space keep working."""
import numpy as np


class _LocalBox(object):
    def __init__(self, low, high, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.low = np.asarray(low, dtype=self.dtype)
        self.high = np.asarray(high, dtype=self.dtype)
        self.shape = self.low.shape

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low.min(), self.high.max(), self.shape)


def _pick_box():
    for mod in ("gymnasium.spaces", "gym.spaces"):
        try:
            return __import__(mod, fromlist=["Box"]).Box
        except Exception:
            continue
    return _LocalBox


_Impl = _pick_box()


def Box(low, high, dtype=np.float32):
    low = np.asarray(low, dtype=dtype)
    high = np.asarray(high, dtype=dtype)
    if _Impl is _LocalBox:
        return _LocalBox(low, high, dtype)
    return _Impl(low=low, high=high, dtype=dtype)
