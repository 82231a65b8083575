"""Minimal stand-in for `pettingzoo` used ONLY by tests/golden/gen_golden.py (the package is not installed in this
image).  The reference's SustainDCPettingZooEnv only subclasses ParallelEnv; nothing of pettingzoo's runs on the path
the HARL fixture records."""


class ParallelEnv:
    metadata = {}

    def __init__(self, *a, **k):
        pass

    @property
    def unwrapped(self):
        return self
