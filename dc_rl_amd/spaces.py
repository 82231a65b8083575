"""Observation / action space objects.  gymnasium's `Box` / `Discrete` are used when gymnasium is importable
(the reference depends on gymnasium==0.29.1); otherwise minimal stand-ins with the attributes HARL reads
(`shape`, `n`, `low`, `high`, `dtype`, `__class__.__name__` -- harl/utils/envs_tools.py:16-46)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box, Discrete  # type: ignore
    from gymnasium import Env  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium is not installed in the target image
    HAVE_GYMNASIUM = False

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

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class Env:
        metadata = {}

        def close(self):
            pass
