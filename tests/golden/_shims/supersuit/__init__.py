"""Minimal stand-in for `supersuit` used ONLY by tests/golden/gen_golden.py (not installed in this image).

Implements the two wrappers harlsustaindc_env.py:25-26 applies, with SuperSuit's documented semantics:
  * pad_action_space_v0: pads Discrete action spaces to the largest n -- the three SustainDC agents all have
    Discrete(3), so this is the identity;
  * pad_observations_v0: pads every agent's Box observation with ZEROS at the end up to the largest observation
    shape (26 here) and reports the padded spaces.
"""
import numpy as np
from gymnasium import spaces


def pad_action_space_v0(env):
    ns = {a: s.n for a, s in env.action_spaces.items()}
    assert len(set(ns.values())) == 1, "the shim only covers equal Discrete spaces (identity padding)"
    return env


class _PadObs:
    def __init__(self, env):
        self._env = env
        self.possible_agents = env.possible_agents
        self.agents = env.agents
        self.action_spaces = env.action_spaces
        self._dim = max(s.shape[0] for s in env.observation_spaces.values())
        self.observation_spaces = {a: spaces.Box(low=np.float32(-2.0), high=np.float32(2.0), shape=(self._dim,), dtype=np.float32)
                                   for a in env.observation_spaces}

    @property
    def unwrapped(self):
        return self._env.unwrapped

    def _pad(self, obs):
        out = {}
        for a, o in obs.items():
            o = np.asarray(o)
            p = np.zeros(self._dim, dtype=o.dtype)
            p[:o.shape[0]] = o
            out[a] = p
        return out

    def reset(self, seed=None, options=None):
        return self._pad(self._env.reset(seed=seed, options=options))

    def step(self, actions):
        obs, rew, term, trunc, info = self._env.step(actions)
        return self._pad(obs), rew, term, trunc, info

    def render(self, *a, **k):
        return self._env.render(*a, **k)

    def close(self):
        return self._env.close()


def pad_observations_v0(env):
    return _PadObs(env)
