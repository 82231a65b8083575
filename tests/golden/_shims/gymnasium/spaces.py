"""Declarative-space stubs (see package docstring)."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is None:
            shape = low.shape
        self.shape = tuple(shape)
        self.low = np.broadcast_to(low, self.shape).copy()
        self.high = np.broadcast_to(high, self.shape).copy()
        self.dtype = np.dtype(dtype)


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
