"""`SustainDC(env_config)`: the reference's single-environment Gymnasium-style surface (sustaindc_env.py:90-737)
over the HIP engine (a batch of one).  Same constructor keys (`EnvConfig.DEFAULT_CONFIG`), same `reset()` /
`step()` return structure, same attribute names HARL's adapters read.  For throughput use `SustainDCVecEnv`.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .agents import BaseBatteryAgent, BaseHVACAgent, BaseLoadShiftingAgent
from .spaces import Env
from .vec_env import AGENTS, DEFAULT_ENV_ARGS, OBS_DIMS, SustainDCVecEnv


class EnvConfig(dict):
    """sustaindc_env.py:34-87"""

    DEFAULT_CONFIG = dict(DEFAULT_ENV_ARGS)

    def __init__(self, raw_config):
        dict.__init__(self, self.DEFAULT_CONFIG.copy())
        for key, val in raw_config.items():
            self[key] = val


class SustainDC(Env):
    def __init__(self, env_config, device: int = 0, seed: int = 1):
        env_config = EnvConfig(env_config)
        self.env_config = env_config
        self.agents = list(env_config["agents"])
        self.location = env_config["location"]
        self.datacenter_capacity_mw = env_config["datacenter_capacity_mw"]
        self.dc_config_file = env_config["dc_config_file"]
        self.timezone_shift = env_config["timezone_shift"]
        self.days_per_episode = env_config["days_per_episode"]
        self.month = env_config.get("month") if env_config.get("month") is not None else 0
        # (the shared-observation option belongs to the HARL layer above this class: whichever, no warning about its absence here)
        vec_args = dict(env_config, _allow_agent_subset=True)
        vec_args.setdefault("nonoverlapping_shared_obs_space", True)
        self._vec = SustainDCVecEnv(vec_args, n_envs=1, seed=seed,
                                    months=[self.month], device=device, auto_reset=False)
        self.ls_env, self.dc_env, self.bat_env = self._vec.ls_env, self._vec.dc_env, self._vec.bat_env
        # spaces of the trained agents only; the others are played by base agents (sustaindc_env.py:172-191)
        sub_envs = {"agent_ls": self.ls_env, "agent_dc": self.dc_env, "agent_bat": self.bat_env}
        self.observation_space = [sub_envs[a].observation_space for a in AGENTS if a in self.agents]
        self.action_space = [sub_envs[a].action_space for a in AGENTS if a in self.agents]
        base = {"agent_ls": BaseLoadShiftingAgent, "agent_dc": BaseHVACAgent, "agent_bat": BaseBatteryAgent}
        self.base_agents = {a: base[a]() for a in AGENTS if a not in self.agents}
        self._agent_ids = set(self.agents)
        self.init_day = self._vec.engine.get_state("day_lo")[0] + 7 if self.month else 0
        self.infos = {}
        self.actions_are_logits = env_config.get("actions_are_logits", False)

    # -- helpers -------------------------------------------------------------------------------------
    def _split(self, obs326):
        return {a: np.array(obs326[k, :OBS_DIMS[k]], dtype=np.float32) for k, a in enumerate(AGENTS) if a in self._agent_ids}

    def seed(self, seed=None):
        self._vec.seed(seed or 1)

    # -- gym surface ---------------------------------------------------------------------------------
    def reset(self):
        """-> {agent: obs}  (old-gym style, sustaindc_env.py:531)"""
        obs, _, _ = self._vec.reset()
        states = self._split(obs[0])
        self.infos = {**{a: {} for a in self.agents}, "__common__": {"states": states}}
        return states

    def step(self, action_dict):
        """-> (obs, rew, terminateds, truncateds, info), dicts keyed by agent (sustaindc_env.py:533-621)"""
        # agents that are not trained are played by the base do-nothing agents (sustaindc_env.py:623-655)
        a = np.array([[int(action_dict[k]) if k in self._agent_ids else int(self.base_agents[k].act()) for k in AGENTS]],
                     dtype=np.int32)
        obs, _, rew, dones, infos, _ = self._vec.step(a)
        terminal = bool(dones[0, 0])
        o = self._split(self._vec.engine.final_obs.cpu().numpy()[0]) if terminal else self._split(obs[0])
        r = {k: float(rew[0, j, 0]) for j, k in enumerate(AGENTS) if k in self._agent_ids}
        terminateds = {k: False for k in self.agents}
        truncateds = {k: terminal for k in self.agents}     # _handle_terminal (sustaindc_env.py:713-718)
        terminateds["__all__"] = False
        truncateds["__all__"] = terminal
        common = dict(infos[0][0])
        info = {k: common for k in AGENTS}                  # _populate_info_dict fills all three (sustaindc_env.py:702-709)
        info["__common__"] = common
        self.infos = {**{k: common for k in self.agents}, "__common__": {"states": o}}
        return o, r, terminateds, truncateds, info

    def render(self, *a, **k):
        return None

    def close(self):
        self._vec.close()

    def get_avail_agent_actions(self, agent_id):
        return [1] * self.action_space[agent_id].n

    def get_avail_actions(self):
        return [self.get_avail_agent_actions(i) for i in range(len(self.action_space))]
