"""Multi-GPU: environment instances are independent, so a job shards the env index range across one process
per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) with no
data-path collective.  The only exchange is the optional all-reduce of episode-return statistics
(SURVEY.md section 8(e)): 7 doubles {sum r[3], sum r^2[3], episodes}, latency-bound."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced env-index range of `rank`: sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group(backend: str = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a documented requirement of this platform, not a tuning choice: its host driver supports dmabuf device-memory IPC only
        # (RCCL's intra-node transports open their peers' buffers with it; without the variable: `hipIpcGetMemHandle: invalid
        # argument`).  The image exports it already; this is for a launcher that dropped it.  Not measurable on a one-GPU lease
        # (a world of one rank opens no peer buffer)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


@dataclass
class ReturnStats:
    """Episode-return statistics accumulated per rank and reduced over the job."""
    sums: "object" = None   # tensor [7] float64: sum r[3], sum r^2[3], episodes

    @classmethod
    def zeros(cls, device="cpu"):
        import torch
        return cls(torch.zeros(7, dtype=torch.float64, device=device))

    def add_episode_returns(self, returns):
        """returns: tensor / array [n_finished, 3] of finished-episode returns on this rank."""
        import torch
        r = torch.as_tensor(returns, dtype=torch.float64, device=self.sums.device).reshape(-1, 3)
        self.sums[0:3] += r.sum(0)
        self.sums[3:6] += (r * r).sum(0)
        self.sums[6] += r.shape[0]

    def all_reduce(self):
        """Sum over ranks (RCCL / gloo); returns a new ReturnStats with the job-wide totals."""
        import torch.distributed as dist
        t = self.sums.clone()
        if dist.is_available() and dist.is_initialized():     # (a world of one still goes through the backend: RCCL / gloo)
            if dist.get_backend() == "nccl" and not t.is_cuda:
                import torch
                t = t.to(torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(t)
                t = t.to(self.sums.device)
            else:
                dist.all_reduce(t)
        return ReturnStats(t)

    def mean_std(self):
        n = max(1.0, float(self.sums[6]))
        mean = (self.sums[0:3] / n).cpu().numpy()
        var = (self.sums[3:6] / n).cpu().numpy() - mean ** 2
        return mean, np.sqrt(np.maximum(var, 0.0)), int(self.sums[6])


def make_sharded_train_env(env_name, seed, n_total_envs, env_args, return_torch=True):
    """This rank's shard of an `n_total_envs`-env job: months follow the GLOBAL env index and the reset RNG is keyed
    on (job seed, global env index, episode) -- `env_index_base` of the shard's engine -- so the job is the same set
    of environments, drawing the same start days / hours / weather noise, whatever the GPU count
    (tests/test_gpu_distributed.py checks a shard against the same envs of the unsharded batch)."""
    from .envs_tools import make_train_env
    rank, local_rank, world = init_process_group()
    lo, hi = shard_range(n_total_envs, rank, world)
    env = make_train_env(env_name, seed, hi - lo, env_args, device=local_rank if world > 1 else 0,
                         return_torch=return_torch, rank_offset=lo)
    return env, (lo, hi)
