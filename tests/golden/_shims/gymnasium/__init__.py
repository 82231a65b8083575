"""Minimal stand-in for `gymnasium` used ONLY by tests/golden/gen_golden.py.

The reference (dc-rl) imports gymnasium for `Env`, `spaces.Box` and
`spaces.Discrete` (declarative only -- no arithmetic on the hot path).
gymnasium is not installed in this image, so the golden-vector generator
registers this stub before importing the reference.  It is test tooling; the
product package has its own `spaces` fallback and never imports this file.
"""
from . import spaces  # noqa: F401


class Env:
    metadata = {}

    def __init__(self, *a, **k):
        pass

    def reset(self, *, seed=None, options=None):
        return None

    def close(self):
        pass
