"""ONE `ShareVecEnv` over several MI355X in ONE process: what an unchanged single-process HARL runner needs to use more than
one GPU.

The reference's `make_train_env` hands the runner one vector env for all `n_threads` environments
(harl/utils/envs_tools.py:49-75, harl/envs/env_wrappers.py:222-297: one worker process per env behind pipes); the runner itself is
a single process.  `dc_rl_amd.distributed` shards a job over one PROCESS per GPU (torch.distributed / RCCL) -- right for a
data-parallel trainer, no use to that runner.  Here the env index range is cut into contiguous ranges, one
`SustainDCVecEnv` (one C-ABI handle, one HIP stream) per device; `step()` hands EVERY device its actions and its launch
before it waits for any of them, the outputs arrive in ONE pinned host block per step (NumPy mode: each device copies its
slice, `infos` is one lazy sequence over the gathered info block) or stay on their devices (`return_torch=True`: tuples of
per-device tensors, in env order).  Months and the reset RNG follow the GLOBAL env index (`env_index_base`), so the job is the
same set of environments whatever the number of devices -- `devices=[0, 0]` (two handles on one GPU) gives the unsharded
batch bit for bit (tests/test_gpu_multi_device.py).  No collective: the return statistics / logger sums of the shards are
added up on the host ("one process, one stream per device")."""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .distributed import shard_range
from .vec_env import LOGGER_KEYS, LazyInfos, ShareVecEnv, SustainDCVecEnv, _FinalObs


def shard_ranges(n_envs: int, n_shards: int):
    """Contiguous, balanced env-index ranges with EVEN boundaries (the kernels specialised for the common case step env pairs
    / quads; only the last range may be odd-sized); empty ranges are dropped."""
    if n_shards < 1 or n_envs < 1:
        raise ValueError("n_envs and the number of devices must be positive")
    bounds = [min(n_envs, 2 * ((n_envs * d + n_shards) // (2 * n_shards))) for d in range(n_shards)] + [n_envs]
    return [(lo, hi) for lo, hi in zip(bounds[:-1], bounds[1:]) if hi > lo]


class _ShardedExtra:
    """`extra` of the gathered infos: (global env, agent) -> the owning shard's entry."""

    def __init__(self, ranges, extras):
        self._ranges, self._extras = ranges, extras

    def get(self, key, default=None):
        e, a = key
        for (lo, hi), ex in zip(self._ranges, self._extras):
            if lo <= e < hi:
                return ex.get((e - lo, a), default) if ex else default
        return default

    def __bool__(self):
        return any(bool(ex) for ex in self._extras)


class SustainDCMultiDeviceVecEnv(ShareVecEnv):
    def __init__(self, env_args=None, n_envs: int = 1, seed: int = 0, months: Optional[Sequence[int]] = None,
                 devices: Sequence[int] = (0,), return_torch: bool = False, auto_reset: bool = True,
                 data_root: Optional[str] = None, env_index_base: int = 0, snapshot_infos: bool = False):
        import torch
        self._torch = torch
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        nd = torch.cuda.device_count()
        for d in devices:
            if not 0 <= d < nd:
                raise ValueError(f"device {d} is not visible (torch.cuda.device_count() = {nd})")
        self.devices = devices
        self.ranges = shard_ranges(n_envs, len(devices))
        self.devices = devices[:len(self.ranges)]
        per_env = list(env_args) if isinstance(env_args, (list, tuple)) else None
        if per_env is not None and len(per_env) != n_envs:
            raise ValueError("env_args list must have n_envs entries")
        months = list(months) if months is not None else None
        self.shards: List[SustainDCVecEnv] = []
        self.streams = []
        for dev, (lo, hi) in zip(self.devices, self.ranges):
            with torch.cuda.device(dev):
                st = torch.cuda.Stream(device=dev)
                with torch.cuda.stream(st):
                    sh = SustainDCVecEnv(per_env[lo:hi] if per_env is not None else env_args, n_envs=hi - lo, seed=seed,
                                         months=months[lo:hi] if months is not None else None, device=dev,
                                         return_torch=return_torch, auto_reset=auto_reset, data_root=data_root,
                                         env_index_base=env_index_base + lo, snapshot_infos=snapshot_infos)
            self.shards.append(sh)
            self.streams.append(st)
        s0 = self.shards[0]
        self.return_torch = return_torch
        self.snapshot_infos = bool(snapshot_infos)
        self.agents, self.n_agents, self._agent_idx = s0.agents, s0.n_agents, s0._agent_idx
        self.obs_width, self.share_concat, self.share_dim = s0.obs_width, s0.share_concat, s0.share_dim
        self.episode_steps, self.policy, self.reward_method = s0.episode_steps, s0.policy, s0.reward_method
        self.months = [m for sh in self.shards for m in sh.months]
        self.data_source = s0.data_source
        self._const = [c for sh in self.shards for c in sh._const]
        ShareVecEnv.__init__(self, n_envs, s0.observation_space, s0.share_observation_space, s0.action_space)
        self._avail_np = np.ones((n_envs, self.n_agents, 3), dtype=np.float32)
        self._host = None
        self._host_flip = 0
        self._gen = 0
        self._need_reset = True

    # ------------------------------------------------------------------ helpers
    def _on(self, d):
        t = self._torch
        return _DevCtx(t, self.devices[d], self.streams[d])

    def _host_sets(self):
        """Two alternating sets of pinned host arrays for the whole job (a step's arrays stay valid until the step after next)."""
        t = self._torch
        if self._host is None:
            N = self.num_envs
            self._host = []
            for _ in range(2):
                full = {"obs": t.empty((N, L.N_AGENTS, L.OBS_PAD), dtype=t.float32, pin_memory=True),
                        "share": t.empty((N, L.SHARE_OBS_DIM), dtype=t.float32, pin_memory=True),
                        "rew": t.empty((N, L.N_AGENTS), dtype=t.float32, pin_memory=True),
                        "info": t.empty((N, L.INFO_DIM), dtype=t.float32, pin_memory=True),
                        "done": t.empty((N,), dtype=t.uint8, pin_memory=True)}
                full["slices"] = [{k: v[lo:hi] for k, v in full.items()} for lo, hi in self.ranges]
                self._host.append(full)
        self._host_flip ^= 1
        return self._host[self._host_flip]

    def _sel(self, x):
        return x if self.n_agents == 3 else x[:, self._agent_idx]

    def _sel_obs(self, obs):
        o = self._sel(obs)
        return o if self.obs_width == L.OBS_PAD else o[:, :, :self.obs_width]

    def seed(self, seed: int):
        for sh in self.shards:
            sh.seed(seed)

    # ------------------------------------------------------------------ ShareVecEnv API
    def reset(self):
        outs = []
        for d, sh in enumerate(self.shards):      # (device-side resets: all enqueued before the first result is read)
            with self._on(d):
                outs.append(sh.engine.reset())
                sh._need_reset = False
        self._need_reset = False
        if self.return_torch:
            res = []
            for d, sh in enumerate(self.shards):
                with self._on(d):
                    obs, share = outs[d]
                    res.append((sh._sel_obs(obs), sh._share3(share, obs), sh._avail))
            self._publish()
            return tuple(r[0] for r in res), tuple(r[1] for r in res), tuple(r[2] for r in res)
        host = []
        for d in range(len(self.shards)):           # (read back ON the shard's stream: it is not ordered against any other)
            with self._on(d):
                host.append((outs[d][0].cpu().numpy(), outs[d][1].cpu().numpy()))
        obs = np.concatenate([o for o, _ in host], axis=0)
        share = np.concatenate([s for _, s in host], axis=0)
        return self._sel_obs(obs), self._share3_np(share, obs), self._avail_np

    def _publish(self):
        """Device-resident outputs are produced on the shards' own streams: make each device's CURRENT stream (where the
        caller's policy runs) wait for them -- a device-side dependency, no host synchronisation."""
        t = self._torch
        for dev, st in zip(self.devices, self.streams):
            t.cuda.current_stream(dev).wait_stream(st)

    def _subscribe(self):
        """... and the shards' streams wait for what the caller's streams have produced so far (its action tensors)."""
        t = self._torch
        for dev, st in zip(self.devices, self.streams):
            st.wait_stream(t.cuda.current_stream(dev))

    def _share3_np(self, share, obs):
        if self.share_concat:
            share = self._sel_obs(obs).reshape(self.num_envs, self.share_dim)
        return np.broadcast_to(share[:, None, :], (self.num_envs, self.n_agents, share.shape[1]))

    def step_async(self, actions):
        """actions: ONE [N, n_agents(, 1)] batch (NumPy array, host tensor, nested list: the ShareVecEnv contract), or -- the
        device-resident form -- a TUPLE of torch tensors, one per shard in env order, each of its shard's [n, n_agents(, 1)] and on
        its shard's device.  Only that exact form is taken as per-device parts (a plain list of per-env rows never is, whatever its
        length); `return_torch=True` correspondingly RETURNS per-device tuples (see step_wait), not [N, ...] arrays."""
        t = self._torch
        per_device = (isinstance(actions, tuple) and len(actions) == len(self.shards) and
                      all(isinstance(p, t.Tensor) for p in actions))
        if per_device:
            for p, (lo, hi), sh in zip(actions, self.ranges, self.shards):
                if p.shape[0] != hi - lo or p.device != sh.engine.device:
                    raise ValueError(f"per-device actions: shard [{lo}, {hi}) on {sh.engine.device} was handed a tensor of "
                                     f"{tuple(p.shape)} on {p.device}")
            parts = list(actions)
        else:
            a = actions.reshape(self.num_envs, self.n_agents) if hasattr(actions, "reshape") else \
                np.asarray(actions).reshape(self.num_envs, self.n_agents)
            parts = [a[lo:hi] for lo, hi in self.ranges]
        if any(hasattr(p, "is_cuda") and p.is_cuda for p in parts):
            self._subscribe()
        for d, sh in enumerate(self.shards):
            with self._on(d):
                sh.step_async(parts[d])
        self._act_parts = parts

    def step_wait(self):
        if self._need_reset:
            raise RuntimeError("call reset() before step()")
        t = self._torch
        k = self.n_agents
        if self.return_torch:
            # device-resident: every shard's step_wait only enqueues (no host synchronisation on that path)
            res = []
            for d, sh in enumerate(self.shards):
                with self._on(d):
                    res.append(sh.step_wait())
            self._publish()
            return tuple(tuple(r[j] for r in res) for j in range(6))
        hs = self._host_sets()
        applied = []
        for d, sh in enumerate(self.shards):          # every device gets its launch and its copies ...
            with self._on(d):
                applied.append(sh._launch_into(hs["slices"][d]))
        for st in self.streams:                       # ... before any is waited for
            st.synchronize()
        done_h = hs["done"].numpy().astype(bool)
        extras = [None] * len(self.shards)
        if done_h.any():
            for d, (sh, (lo, hi)) in enumerate(zip(self.shards, self.ranges)):
                if done_h[lo:hi].any():
                    with self._on(d):
                        extras[d] = _FinalObs(sh.engine.final_obs.cpu().numpy(), done_h[lo:hi], self._agent_idx, k,
                                              self.share_concat, self.obs_width)
        extra = _ShardedExtra(self.ranges, extras) if any(e is not None for e in extras) else {}
        self._gen += 1
        acts = _ShardedActions(applied)
        infos = LazyInfos(hs["info"], acts, done_h, self._const, extra, self, 1, k, self.shards[0]._info_keys)
        if self.snapshot_infos:
            infos.rows()
        obs = hs["obs"].numpy()
        share = hs["share"].numpy()
        return (self._sel_obs(obs), self._share3_np(share, obs), self._sel(hs["rew"].numpy())[..., None],
                np.repeat(done_h[:, None], k, axis=1), infos, self._avail_np)

    # ------------------------------------------------------------------ reductions over the shards
    def accumulate_logger_sums(self, keys: Sequence[str] = LOGGER_KEYS, enable: bool = True):
        for d, sh in enumerate(self.shards):
            with self._on(d):
                sh.accumulate_logger_sums(keys, enable)

    def read_logger_sums(self, reset: bool = True):
        tot, n = None, 0
        for d, sh in enumerate(self.shards):
            with self._on(d):
                s, n = sh.read_logger_sums(reset)
            tot = s if tot is None else {k: tot[k] + v for k, v in s.items()}
        return tot, n

    def episode_return_sums(self):
        """{sum r[3], sum r^2[3], episodes} of the episodes that ended in the LAST step, added up over the shards on the host
        (what the one-process-per-GPU path all-reduces: dc_rl_amd.distributed.ReturnStats)."""
        from .distributed import ReturnStats
        st = ReturnStats.zeros()
        for d, sh in enumerate(self.shards):
            with self._on(d):
                ld = sh.engine.last_done()
                if ld is not None and ld.any():
                    r = sh.engine.info[:, [L.INFO_IDX["ep_return_ls"], L.INFO_IDX["ep_return_dc"], L.INFO_IDX["ep_return_bat"]]]
                    st.add_episode_returns(r[self._torch.as_tensor(ld, device=r.device)].double().cpu())
        return st

    def close_extras(self):
        for sh in self.shards:
            sh.close()


class _DevCtx:
    """`with` block: device d current, its stream current."""

    def __init__(self, torch, dev, stream):
        self._a = torch.cuda.device(dev)
        self._b = torch.cuda.stream(stream)

    def __enter__(self):
        self._a.__enter__()
        self._b.__enter__()

    def __exit__(self, *exc):
        self._b.__exit__(*exc)
        self._a.__exit__(*exc)


class _ShardedActions:
    """The [N, 3] actions a step applied, as the shards hold them (device tensors on several devices): one host copy when
    `infos[i][a]["ls_action"]` is first read."""

    def __init__(self, parts):
        self._parts = parts
        self._version = None

    def detach(self):
        return self

    def cpu(self):
        return self

    def numpy(self):
        return np.concatenate([p.detach().cpu().numpy() for p in self._parts], axis=0)
