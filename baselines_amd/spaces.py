"""Minimal observation/action space types (gym is not a dependency).

Any object with the same attributes works: spaces are duck-typed — a space with an
integer ``n`` is discrete, one with ``shape``/``dtype`` is a box — which is how
``make_pdtype`` tells Box from Discrete (baselines/common/distributions.py:238-250).
"""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low)
            shape = low.shape
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, self.dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, self.dtype), self.shape)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    def __repr__(self):
        return "Discrete(%d)" % self.n


def is_discrete(space):
    return hasattr(space, "n") and not hasattr(space, "nvec")
