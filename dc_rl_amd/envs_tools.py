"""`make_train_env` / `make_eval_env`: the one function of the HARL stack that is re-targeted
(harl/utils/envs_tools.py:49-103): `n_threads` environments become one device batch instead of
`n_threads` worker processes.  Same month assignment as the reference (month = rank % 12 for rank < 12 else
rank % 3 + 5 unless `month` is given).  Seeding: the reference gives env `rank` the seed `seed + rank * 1000`
(envs_tools.py:63); here the batch has ONE seed and the counter-based reset RNG is keyed on (seed, global env index,
episode) -- distinct, reproducible streams per env, and the same streams for global env i whatever the number of GPUs
the job is sharded over (`rank_offset`).
"""
from __future__ import annotations

from typing import List

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


def get_num_agents(env, env_args, envs):
    """harl/utils/envs_tools.py:147"""
    if env == "sustaindc":
        return envs.n_agents
    raise ValueError(f"Unsupported environment type: '{env}'. Check the environment name and try again.")
