"""`make_train_env` / `make_eval_env`: the one function of the HARL stack that is re-targeted
(harl/utils/envs_tools.py:49-103): `n_threads` environments become one device batch instead of
`n_threads` worker processes.  Same month assignment as the reference (month = rank % 12 for rank < 12 else
rank % 3 + 5 unless `month` is given).  Seeding: the reference gives env `rank` the seed `seed + rank * 1000`
(envs_tools.py:63); here the batch has ONE seed and the counter-based reset RNG is keyed on (seed, global env index,
episode) -- distinct, reproducible streams per env, and the same streams for global env i whatever the number of GPUs
the job is sharded over (`rank_offset`).
"""
from __future__ import annotations

import datetime
from typing import List

import numpy as np

from .vec_env import SustainDCVecEnv


def months_for_ranks(n: int, env_args: dict, rank_offset: int = 0) -> List[int]:
    """harl/utils/envs_tools.py:56-62"""
    out = []
    for rank in range(rank_offset, rank_offset + n):
        if "month" in env_args and env_args["month"] is not None:
            out.append(int(env_args["month"]))
        elif rank < 12:
            out.append(rank % 12)
        else:
            out.append(rank % 3 + 5)   # 33 % June (5), July (6), August (7)
    return out


def make_train_env(env_name, seed, n_threads, env_args, device: int = 0, return_torch: bool = False,
                   rank_offset: int = 0, devices=None):
    """Counterpart of harl/utils/envs_tools.py:49.  `rank_offset` = index of this shard's first env when the
    batch is one shard of a multi-GPU job (see dc_rl_amd.distributed).  `devices=[0, 1, ...]`: ONE vector env over several
    GPUs in this process (dc_rl_amd.multi_device: one handle and one stream per device, contiguous env ranges) -- what a
    single-process HARL runner uses to drive more than one MI355X."""
    if env_name != "sustaindc":
        print("Can not support the " + env_name + "environment.")
        raise NotImplementedError
    if devices is not None:
        from .multi_device import SustainDCMultiDeviceVecEnv
        return SustainDCMultiDeviceVecEnv(env_args, n_envs=n_threads, seed=seed,
                                          months=months_for_ranks(n_threads, env_args, rank_offset), devices=devices,
                                          return_torch=return_torch, env_index_base=rank_offset)
    return SustainDCVecEnv(env_args, n_envs=n_threads, seed=seed,
                           months=months_for_ranks(n_threads, env_args, rank_offset), device=device,
                           return_torch=return_torch, env_index_base=rank_offset)


def make_eval_env(env_name, seed, n_threads, env_args, device: int = 0, return_torch: bool = False,
                  rank_offset: int = 0):
    """Counterpart of harl/utils/envs_tools.py:77 (seed * 50000 + rank * 10000)."""
    if env_name != "sustaindc":
        print("Can not support the " + env_name + "environment.")
        raise NotImplementedError
    return SustainDCVecEnv(env_args, n_envs=n_threads, seed=seed * 50000,
                           months=months_for_ranks(n_threads, env_args, rank_offset), device=device,
                           return_torch=return_torch, env_index_base=rank_offset)


class HARLRenderEnv:
    """What `make_render_env` hands the runner's `render()` (harl/runners/on_policy_base_runner.py:746-852): ONE un-batched HARL env
    with the reference's `HARLSustainDCEnv` surface (harl/envs/sustaindc/harlsustaindc_env.py:10-215) -- per-agent lists instead of
    [N, ...] arrays, no auto-reset (the runner calls `reset()` itself when `dones[0]` comes back true), `render_episode` /
    `experiment_datetime` for its CSV dump -- over a one-env device batch."""

    def __init__(self, env_args, seed: int = 0, device: int = 0):
        self.env_args = env_args
        month = env_args.get("month")
        self._vec = SustainDCVecEnv(env_args, n_envs=1, seed=seed, months=[0 if month is None else int(month)], device=device,
                                    return_torch=False, auto_reset=False, snapshot_infos=True)
        self.n_agents = self._vec.n_agents
        self.agents = list(self._vec.agents)
        self.max_cycles = 25
        self.cur_step = 0
        self.share_observation_space = self._vec.share_observation_space
        self.observation_space = self._vec.observation_space
        self.action_space = self._vec.action_space
        self.discrete = True
        self.is_render = bool(env_args.get("is_render", False))
        if self.is_render:
            self.experiment_datetime = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S")
        self.render_episode = 0
        self._seed = int(seed)

    def reset(self):
        """-> (obs: list[n_agents] of float32[width], shared obs: list[n_agents], available actions) (harlsustaindc_env.py:90-104)"""
        self.render_episode += 1
        self._seed += 1
        self.cur_step = 0
        obs, share, _ = self._vec.reset()
        return [np.array(o) for o in obs[0]], [np.array(s) for s in share[0]], self.get_avail_actions()

    def step(self, actions):
        """actions: [n_agents] or [n_agents, 1] -> (obs, shared obs, [[r]] per agent, dones per agent, info dict per agent,
        available actions) (harlsustaindc_env.py:106-131)"""
        a = np.asarray(actions).reshape(1, self.n_agents)
        obs, share, rew, done, infos, _ = self._vec.step(a)
        self.cur_step += 1
        return ([np.array(o) for o in obs[0]], [np.array(s) for s in share[0]], [[float(r[0])] for r in rew[0]],
                [bool(d) for d in done[0]], [infos[0][k] for k in range(self.n_agents)], self.get_avail_actions())

    def seed(self, seed):
        self._seed = seed
        self._vec.seed(int(seed))

    def get_avail_actions(self):
        return [self.get_avail_agent_actions(i) for i in range(self.n_agents)]

    def get_avail_agent_actions(self, agent_id):
        return [1] * self.action_space[agent_id].n

    def render(self):
        pass    # (the reference's render() is a no-op too: sustaindc_env.py, `render`)

    def close(self):
        self._vec.close()


def make_render_env(env_name, seed, env_args, device: int = 0):
    """Counterpart of harl/utils/envs_tools.py:106-133: the reference's 5-tuple (env, manual_render, manual_expand_dims,
    manual_delay, env_num) around ONE un-batched env seeded `seed * 60000`.  (The reference reads an undefined `rank` when
    `env_args` has no `month` -- a NameError; here that case takes rank 0's month, 0.)"""
    manual_render = True        # manually call the render() function
    manual_expand_dims = True   # manually expand the num_of_parallel_envs dimension
    manual_delay = True         # manually delay the rendering by time.sleep()
    env_num = 1                 # number of parallel envs
    if env_name != "sustaindc":
        print("Can not support the " + env_name + "environment.")
        raise NotImplementedError
    if env_args.get("month") is None:
        env_args["month"] = 0
    env_args["is_render"] = True
    print("Rendering the environment with month: ", env_args["month"])
    env = HARLRenderEnv(env_args, seed=seed * 60000, device=device)
    return env, manual_render, manual_expand_dims, manual_delay, env_num


def get_num_agents(env, env_args, envs):
    """harl/utils/envs_tools.py:147"""
    if env == "sustaindc":
        return envs.n_agents
    raise ValueError(f"Unsupported environment type: '{env}'. Check the environment name and try again.")


# ----------------------------------------------------------------------------------------------------------------------
# ZERO-EDIT DROP-IN.  Both base runners bind the factories by name at import time
# (`from harl.utils.envs_tools import make_eval_env, make_train_env, make_render_env, set_seed, get_num_agents`:
# harl/runners/on_policy_base_runner.py:17-23, off_policy_base_runner.py:10-16) and call them at :103, :110, :127 / :77-86.
# `install_into_harl()` rebinds those names in `harl.utils.envs_tools` -- and in any runner module that was imported
# earlier -- so that an unchanged HARL tree (train_sustaindc.py, the runners, the logger) builds its envs here:
#     import dc_rl_amd; dc_rl_amd.install_into_harl()      # before (or after) `import harl.runners`
_HARL_NAMES = ("make_train_env", "make_eval_env", "make_render_env", "get_num_agents")
_HARL_CLIENTS = ("harl.runners.on_policy_base_runner", "harl.runners.off_policy_base_runner")


def install_into_harl(device=None, return_torch=None, devices=None):
    """Rebind `harl.utils.envs_tools.{make_train_env, make_eval_env, make_render_env, get_num_agents}` to this module's.
    With no arguments the very functions of this module are bound (`harl.utils.envs_tools.make_train_env is
    dc_rl_amd.envs_tools.make_train_env`); `device` / `return_torch` / `devices` bind wrappers with those keyword arguments
    filled in (the runners only ever pass the reference's four positional arguments).  Returns the names bound per module."""
    import functools
    import importlib
    import sys
    et = importlib.import_module("harl.utils.envs_tools")
    kw = {k: v for k, v in (("device", device), ("return_torch", return_torch)) if v is not None}
    new = {"make_train_env": make_train_env, "make_eval_env": make_eval_env, "make_render_env": make_render_env,
           "get_num_agents": get_num_agents}
    if kw or devices is not None:
        tk = dict(kw, **({"devices": devices} if devices is not None else {}))
        new["make_train_env"] = functools.wraps(make_train_env)(functools.partial(make_train_env, **tk))
        new["make_eval_env"] = functools.wraps(make_eval_env)(functools.partial(make_eval_env, **kw))
        if device is not None:
            new["make_render_env"] = functools.wraps(make_render_env)(functools.partial(make_render_env, device=device))
    bound = {}
    for modname in ("harl.utils.envs_tools",) + _HARL_CLIENTS:
        m = et if modname == "harl.utils.envs_tools" else sys.modules.get(modname)
        if m is None:
            continue
        for name in _HARL_NAMES:
            if modname == "harl.utils.envs_tools" or hasattr(m, name):
                if not hasattr(m, "_sdc_original_" + name) and hasattr(m, name):
                    setattr(m, "_sdc_original_" + name, getattr(m, name))
                setattr(m, name, new[name])
                bound.setdefault(modname, []).append(name)
    return bound


def uninstall_from_harl():
    """Undo `install_into_harl()` (tests)."""
    import sys
    for modname in ("harl.utils.envs_tools",) + _HARL_CLIENTS:
        m = sys.modules.get(modname)
        if m is None:
            continue
        for name in _HARL_NAMES:
            if hasattr(m, "_sdc_original_" + name):
                setattr(m, name, getattr(m, "_sdc_original_" + name))
                delattr(m, "_sdc_original_" + name)
