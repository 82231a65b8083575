"""The vector-env boundary HARL runners call (`ShareVecEnv`, harl/envs/env_wrappers.py:53-165), backed by one
MI355X batch instead of one OS process per environment.

`SustainDCVecEnv(env_args, n_envs, ...)` presents N coupled SustainDC environments:
  reset() -> (obs [N,3,26] f32, share_obs [N,3,29] f32, available_actions [N,3,3])        (env_wrappers.py:275-280)
  step(actions [N,3,1] | [N,3]) -> (obs, share_obs, rews [N,3,1], dones [N,3] bool, infos, available_actions)
                                                                                          (env_wrappers.py:262-273)
with the reference's auto-reset semantics (env_wrappers.py:176-190): an env whose episode ends is reset inside
the same step call, the returned obs are the reset obs, and `infos[i][0]["original_obs" | "original_state" |
"original_avail_actions"]` hold the pre-reset values.

Outputs are NumPy arrays by default (what `ShareSubprocVecEnv` returns after `np.stack`); with
`return_torch=True` they are device tensors (views of the engine's buffers, valid until the next call) so that a
GPU policy never round-trips through the host.  `infos` is a lazy sequence: the N x 3 dicts of the reference are
materialised only for the entries a caller touches; an `infos` object read after its info block has been overwritten
raises (never another step's values), `snapshot_infos=True` gives every step's `infos` its own copy;
`accumulate_logger_sums()` / `read_logger_sums()` keep the logger's per-step sums
(harl/envs/sustaindc/sustaindc_logger.py:87-101) in a device-side accumulator that is read once per episode.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections.abc import Mapping, Sequence
from typing import List, Optional, Union

import numpy as np

from . import _lib as L
from . import dc_config, traces
from .engine import SdcEngine
from .make_envs_pyenv import make_bat_fwd_env, make_dc_pyeplus_env, make_ls_env
from .spaces import Box, Discrete

AGENTS = ["agent_ls", "agent_dc", "agent_bat"]
OBS_DIMS = (26, 14, 13)

# the 10 info keys the SustainDC logger sums every step (sustaindc_logger.py:87-101)
LOGGER_KEYS = ["bat_total_energy_with_battery_KWh", "bat_CO2_footprint", "dc_water_usage",
               "ls_unasigned_day_load_left", "ls_tasks_in_queue", "ls_tasks_dropped", "dc_ITE_total_power_kW",
               "dc_CT_total_power_kW", "dc_Compressor_total_power_kW", "dc_HVAC_total_power_kW"]

DEFAULT_ENV_ARGS = {  # EnvConfig.DEFAULT_CONFIG (sustaindc_env.py:38-80)
    "agents": list(AGENTS), "location": "ny", "cintensity_file": "NYIS_NG_&_avgCI.csv",
    "weather_file": "USA_NY_New.York-Kennedy.epw", "workload_file": "Alibaba_CPU_Data_Hourly_1.csv",
    "datacenter_capacity_mw": 1, "timezone_shift": 0, "days_per_episode": 7, "max_bat_cap_Mw": 2,
    "dc_config_file": "dc_config.json", "individual_reward_weight": 0.8, "flexible_load": 0.1,
    "ls_reward": "default_ls_reward", "dc_reward": "default_dc_reward", "bat_reward": "default_bat_reward",
    "evaluation": False, "actions_are_logits": False,
}


class ShareVecEnv(ABC):
    """Abstract vectorised multi-agent env (same surface as harl/envs/env_wrappers.py:53)."""

    closed = False
    viewer = None
    metadata = {"render.modes": []}

    def __init__(self, num_envs, observation_space, share_observation_space, action_space):
        self.num_envs = num_envs
        self.observation_space = observation_space
        self.share_observation_space = share_observation_space
        self.action_space = action_space

    @abstractmethod
    def reset(self):
        pass

    @abstractmethod
    def step_async(self, actions):
        pass

    @abstractmethod
    def step_wait(self):
        pass

    def close_extras(self):
        pass

    def close(self):
        if self.closed:
            return
        self.close_extras()
        self.closed = True

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    @property
    def unwrapped(self):
        return self


class _FinalObs:
    """The `original_obs / original_state / original_avail_actions` entries of the envs that finished in a step
    (env_wrappers.py:176-190 puts them into agent 0's info), built per env on first access from one host copy of the
    step's pre-reset observations."""

    def __init__(self, final_obs: np.ndarray, done: np.ndarray, agent_idx, n_agents: int, concat: bool = False,
                 width: int = L.OBS_PAD):
        self._fo, self._done, self._idx, self._k, self._concat, self._w = final_obs, done, agent_idx, n_agents, concat, width
        self._made = {}

    def get(self, key, default=None):
        e, a = key
        if a != 0 or not self._done[e]:
            return default
        d = self._made.get(e)
        if d is None:
            fo = self._fo[e]
            # states[2][-1] of the PADDED obs: agent_bat's zero padding (harlsustaindc_env.py:25-26, :80)
            if self._concat:    # harlsustaindc_env.py:83-85: the trained agents' padded observations, concatenated
                raw = fo[self._idx][:, :self._w].reshape(-1)
            else:
                raw = np.concatenate([fo[0, :26], fo[1, 11:12], fo[1, 13:14], fo[2, 25:26]])
            d = self._made[e] = {"original_obs": fo[self._idx][:, :self._w].copy(),
                                 "original_state": np.repeat(raw[None, :], self._k, axis=0),
                                 "original_avail_actions": np.ones((self._k, 3), dtype=np.float32)}
        return d


_INFOS = L.load_infos()        # csrc/sdc_infos.c: InfoSeq (`infos`), InfoView (`infos[i][a]`) -- C types, see there
Mapping.register(_INFOS.InfoView)
Sequence.register(_INFOS.InfoRow)
_DERIVED_KEYS = ("ls_action", "bat_a_t", "isterminal")
_KEY_TEMPLATES = {}


def _base_keys(const_keys):
    """The keys of an info dict without extra entries, in iteration order, as ONE shared dict_keys object (a C-level `in`)."""
    t = _KEY_TEMPLATES.get(const_keys)
    if t is None:
        t = _KEY_TEMPLATES[const_keys] = dict.fromkeys(list(L.INFO_COLS) + ["ls_task_age_histogram"] + list(const_keys) +
                                                       list(_DERIVED_KEYS))
    return t.keys()


class _InfoSource:
    """What the views of one step read from: the step's [N, K] info block (fetched once, guarded), and the entries that are
    not a column of it.  Referenced by the C side (InfoCore); holds no reference back to the `infos` object."""

    __slots__ = ("_t", "_rows", "actions", "_act_version", "_act_host", "done", "const", "extra", "_owner", "_gen", "_valid_for",
                 "_keys")

    def __init__(self, info_tensor, actions, done, const, extra, owner, valid_for, keys):
        self._t = info_tensor
        self._rows = None
        self.actions = actions
        # the [N, 3] action tensor may be the caller's own (a GPU policy's output, taken without a copy): remember its
        # version so that an in-place overwrite is noticed instead of read as this step's actions
        self._act_version = getattr(actions, "_version", None)
        self._act_host = None
        self.done = done
        self.const = const
        self.extra = extra
        self._owner, self._gen, self._valid_for = owner, (owner._gen if owner is not None else 0), valid_for
        self._keys = keys

    def applied_actions(self):
        """[N, 3] host copy of the actions this step applied (read once, on first access)."""
        if self._act_host is None:
            a = self.actions
            if self._act_version is not None and a._version != self._act_version:
                raise RuntimeError("the action tensor passed to step() has been modified in place since; read "
                                   "infos[...]['ls_action'] before overwriting it, or pass step() a copy")
            self._act_host = a.detach().cpu().numpy().copy() if hasattr(a, "detach") else np.array(a)
        return self._act_host

    def rows(self):
        """The step's info block as ONE float32 [N, K] host array: one bulk conversion on first access."""
        if self._rows is None:
            # the block this object reads from is reused by later steps: never hand out another step's values
            if self._owner is not None and self._owner._gen - self._gen > self._valid_for:
                raise RuntimeError("these infos belong to an earlier step whose info block has been overwritten; read them "
                                   "before stepping again or construct the env with snapshot_infos=True")
            # (a copy: the pinned host buffers are reused two steps later)
            r = self._t.detach().cpu().numpy() if hasattr(self._t, "detach") else np.asarray(self._t)
            self._rows = np.array(r, dtype=np.float32, order="C", copy=True)
            self._owner = None
        return self._rows

    # ---- entries that are not a column of the block or a per-env constant (the C side asks for these) ----
    def _slow_get(self, env, agent, key):
        ex = self.extra.get((env, agent))
        if ex and key in ex:
            return ex[key]
        if key == "ls_task_age_histogram":
            j = L.INFO_IDX["ls_task_age_hist0"]
            return np.array(self.rows()[env, j:j + 5], dtype=np.float64)
        if key == "ls_action":
            return int(self.applied_actions()[env, 0])
        if key == "bat_a_t":
            return ("charge", "discharge", "idle")[int(self.rows()[env, L.INFO_IDX["bat_action"]])]
        if key == "isterminal":
            return bool(self.done[env])
        raise KeyError(key)

    def _has_extra(self, env, agent):
        return bool(self.extra.get((env, agent)))

    def _full_keys(self, env, agent):
        return list(self._keys) + list(self.extra.get((env, agent)) or ())


class LazyInfos(_INFOS.InfoSeq):
    """`infos` of a step: tuple[N] of list[3] of dict in the reference (env_wrappers.py:262-273); here a C sequence over one
    [N, K] array (csrc/sdc_infos.c): `infos[i]` is the env's cached row of per-agent views (a read-only sequence), `infos[i][a]` a read-only
    mapping whose `get` / `[]` / `in` / `keys()` run in C -- so the access pattern of an unchanged HARL runner
    (`infos[i][0].get(key, 0)` for 10 keys per env in sustaindc_logger.py:87-101, `"bad_transition" in info[0].keys()` per
    env in on_policy_base_runner.py:459-471) costs what it costs on plain dicts, without building N x 3 dicts per step."""

    def __init__(self, info_tensor, actions, done, const, extra, owner=None, valid_for=0, n_agents=3, keys=None):
        const = const if isinstance(const, list) else list(const)
        if keys is None:      # (the envs hand their own in: one lookup per env object, not per step)
            keys = _base_keys(tuple(const[0].keys()) if const else ())
        src = _InfoSource(info_tensor, actions, done, const, extra, owner, valid_for, keys)
        super().__init__(len(done), n_agents, L.INFO_IDX, keys, const, src, bool(extra))
        self._src = src

    actions = property(lambda self: self._src.actions)
    done = property(lambda self: self._src.done)
    const = property(lambda self: self._src.const)
    extra = property(lambda self: self._src.extra)

    def applied_actions(self):
        return self._src.applied_actions()

    def rows(self):
        return self._src.rows()


Sequence.register(LazyInfos)


_WARNED_SHARE_DEFAULT = False


def _merge_args(env_args: Optional[dict]) -> dict:
    a = dict(DEFAULT_ENV_ARGS)
    if env_args:
        a.update(env_args)
    L.reward_codes(a)   # NotImplementedError for a reward method the device does not run
    # options of the reference's HARL layer that change what the runner receives: never silently ignored.
    # nonoverlapping_shared_obs_space: harlsustaindc_env.py:53 reads it with .get(..., False) -- an absent key means the
    # CONCATENATED shared observation (3 x 26 floats, :83-85), True (the shipped YAML) the 29-float layout (:78-80).
    # (sustaindc_ptzoo.py:29 subscripts the key, so the reference itself raises KeyError when it is absent; the HARL
    # layer's default is the one taken here.)
    if "nonoverlapping_shared_obs_space" not in a:
        global _WARNED_SHARE_DEFAULT
        if not _WARNED_SHARE_DEFAULT:
            _WARNED_SHARE_DEFAULT = True
            import warnings
            warnings.warn("env_args has no 'nonoverlapping_shared_obs_space': taking the HARL layer's .get default (False = the "
                          "trained agents' padded observations concatenated, 78 floats for three agents); the reference's shipped "
                          "sustaindc.yaml sets True (the 29-float layout)", stacklevel=3)
    a["nonoverlapping_shared_obs_space"] = bool(a.get("nonoverlapping_shared_obs_space", False))
    if not a.get("partial_obs", True):
        raise NotImplementedError("Fully observable states are no longer supported. Please set 'partial_obs' to True.")
    # actions_are_logits: the reference stores the flag and never reads it (sustaindc_env.py:204-205): accepted, ignored
    unknown = [x for x in a["agents"] if x not in AGENTS]
    if unknown:
        raise ValueError(f"unknown agents {unknown}; the environment has {AGENTS}")
    if not a["agents"]:
        raise ValueError("at least one agent must be trained")
    return a


class SustainDCVecEnv(ShareVecEnv):
    def __init__(self, env_args: Union[dict, List[dict], None] = None, n_envs: int = 1, seed: int = 0,
                 months: Optional[Sequence[int]] = None, device: int = 0, return_torch: bool = False,
                 auto_reset: bool = True, data_root: Optional[str] = None, env_index_base: int = 0,
                 snapshot_infos: bool = False):
        per_env = [_merge_args(a) for a in env_args] if isinstance(env_args, (list, tuple)) else [_merge_args(env_args)] * n_envs
        if len(per_env) != n_envs:
            raise ValueError("env_args list must have n_envs entries")
        days = {a["days_per_episode"] for a in per_env}
        if len(days) != 1:
            raise ValueError("all envs of one batch must share days_per_episode")
        rcodes = {L.reward_codes(a) for a in per_env}
        if len(rcodes) != 1:
            raise ValueError("all envs of one batch must share ls_reward / dc_reward / bat_reward")
        self.reward_method = rcodes.pop()
        # A subset of agents (sustaindc_env.py:172-191, :623-655): the others are played by the reference's base
        # do-nothing agents -- on the device (sdc_config.policy = DO_NOTHING for their slots) -- and the surface carries
        # the trained agents only, in the reference's order.  (`_allow_agent_subset`: the single-env facade SustainDC
        # keeps the three-agent arrays and plays the base agents itself.)
        subsets = {tuple(x for x in AGENTS if x in a["agents"]) for a in per_env}
        if len(subsets) != 1:
            raise ValueError("all envs of one batch must train the same agents")
        full = bool(per_env[0].get("_allow_agent_subset"))
        share_modes = {a["nonoverlapping_shared_obs_space"] for a in per_env}
        if len(share_modes) != 1:
            raise ValueError("all envs of one batch must share nonoverlapping_shared_obs_space")
        self.share_concat = not share_modes.pop()
        self.agents = list(AGENTS) if full else list(subsets.pop())
        self._agent_idx = [AGENTS.index(x) for x in self.agents]
        self.n_agents = len(self.agents)
        self.policy = tuple(0 if (full or x in self.agents) else 1 for x in AGENTS)
        self.return_torch = return_torch
        self.episode_steps = int(days.pop()) * 96
        if months is None:
            months = [a.get("month", 0) if a.get("month") is not None else 0 for a in per_env]
        self.months = [int(m) for m in months]
        # unique trace sets and data-centre parameter sets
        loc_keys, cfg_keys, loc_id, cfg_id = [], [], [], []
        for a in per_env:
            lk = (a["location"].lower(), a["workload_file"], int(a["timezone_shift"]))
            ci_loc, _ = traces.obtain_paths(a["location"])
            ck = (a["dc_config_file"], float(a["datacenter_capacity_mw"]), ci_loc)
            if lk not in loc_keys:
                loc_keys.append(lk)
            if ck not in cfg_keys:
                cfg_keys.append(ck)
            loc_id.append(loc_keys.index(lk))
            cfg_id.append(cfg_keys.index(ck))
        self.tables = [traces.get_tables(lk[0], lk[1], lk[2], data_root=data_root) for lk in loc_keys]
        self.data_source = self.tables[0]["source"]
        # the same factory calls SustainDC.__init__ makes (sustaindc_env.py:148-160)
        self.ls_env = make_ls_env(month=self.months[0], n_vars_ci=8, n_vars_energy=0, n_vars_battery=0, queue_max_len=1000)
        self.dc_envs = [make_dc_pyeplus_env(month=self.months[0] + 1, location=ck[2], dc_config_file=ck[0],
                                            datacenter_capacity_mw=ck[1], max_bat_cap_Mw=per_env[0]["max_bat_cap_Mw"],
                                            use_ls_cpu_load=True, add_cpu_usage=False)[0] for ck in cfg_keys]
        self.dc_env = self.dc_envs[0]
        tot = self.dc_env.ranges["Facility Total Electricity Demand Rate(Whole Building)"]
        self.bat_env = make_bat_fwd_env(month=self.months[0], max_bat_cap_Mwh=self.dc_env.ranges["max_battery_energy_Mwh"],
                                        max_dc_pw_MW=tot[1] / 1e6, dcload_max=tot[1], dcload_min=tot[0], n_fwd_steps=8)
        self.bat_env.dcload_max = self.dc_env.power_ub_kW / 4   # sustaindc_env.py:158-160
        self.bat_env.dcload_min = self.dc_env.power_lb_kW / 4
        self.engine = SdcEngine(n_envs, episode_steps=self.episode_steps, device=device, n_locations=len(loc_keys),
                                n_dc_configs=len(cfg_keys), auto_reset=auto_reset, seed=seed, queue_max_len=1000,
                                reward_method=self.reward_method, env_index_base=env_index_base, policy=self.policy)
        for i, tb in enumerate(self.tables):
            self.engine.set_tables(i, tb["W"], tb["C"], tb["T"], tb["WB"])
        for i, e in enumerate(self.dc_envs):
            self.engine.set_dc_params(i, e.sized)
        init_day = np.array([traces.get_init_day(m) for m in self.months])
        self.engine.assign(np.array(loc_id), np.array(cfg_id), np.maximum(0, init_day - 7), np.minimum(364, init_day + 7))
        self._cfg_id = cfg_id
        # per-env constant info entries (envs/dc_gym.py:224-227, envs/bat_env_fwd_view.py:118-121, carbon_ls.py:294-297)
        self._const = []
        for i in range(n_envs):
            e = self.dc_envs[cfg_id[i]]
            c = e.DC_Config
            self._const.append({
                "ls_queue_max_len": 1000, "ls_norm_load_left": 0, "ls_unasigned_day_load_left": 0, "ls_penalty_flag": 0,
                "ls_enforced": 0, "dc_power_lb_kW": e.power_lb_kW, "dc_power_ub_kW": e.power_ub_kW,
                "dc_CW_pump_power_kW": (c.CW_PRESSURE_DROP * c.CW_WATER_FLOW_RATE) / c.CW_PUMP_EFFICIENCY,
                "dc_CT_pump_power_kW": (c.CT_PRESSURE_DROP * c.CT_WATER_FLOW_RATE) / c.CT_PUMP_EFFICIENCY,
                "bat_max_bat_cap": e.sized["bat_capacity"], "bat_dcload_min": e.power_lb_kW / 4,
                "bat_dcload_max": e.power_ub_kW / 4,
            })
        self._info_keys = _base_keys(tuple(self._const[0].keys()))
        # HARL pads every agent to the widest space OF THE TRAINED AGENTS (harlsustaindc_env.py:25-26, :30-33: 26 whenever
        # agent_ls is trained, 14 for dc + bat, 13 for bat alone)
        self.obs_width = L.OBS_PAD if full else max(OBS_DIMS[i] for i in self._agent_idx)
        obs_space = [Box(low=-2.0, high=2.0, shape=(self.obs_width,), dtype=np.float32) for _ in self.agents]
        if self.share_concat:
            # sustaindc_ptzoo.py:32-44: max observation width x number of agents, Box(0, 1)
            self.share_dim = self.obs_width * len(self.agents)
            share_space = [Box(low=np.float32(0), high=np.float32(1), shape=(self.share_dim,), dtype=np.float32) for _ in self.agents]
        else:
            self.share_dim = L.SHARE_OBS_DIM
            share_space = [Box(low=-2.0, high=2.0, shape=(L.SHARE_OBS_DIM,), dtype=np.float32) for _ in self.agents]
        act_space = [Discrete(3) for _ in self.agents]
        ShareVecEnv.__init__(self, n_envs, obs_space, share_space, act_space)
        import torch
        self._torch = torch
        self._avail = torch.ones((n_envs, self.n_agents, 3), dtype=torch.float32, device=self.engine.device)
        self._avail_np = np.ones((n_envs, self.n_agents, 3), dtype=np.float32)
        self._idx_t = torch.as_tensor(self._agent_idx, device=self.engine.device)
        self._logger_acc = None
        self._actions = None
        self._need_reset = True
        self._host = None       # pinned host output buffers (NumPy outputs only)
        self._host_flip = 0
        self._no_done = np.zeros(n_envs, dtype=bool)
        self._gen = 0
        self._torch_views = None
        self.snapshot_infos = bool(snapshot_infos)
        self._act_pin = None    # pinned host staging of NumPy actions
        self._act_flip = 0

    # ------------------------------------------------------------------ helpers
    def _out(self, t):
        return t if self.return_torch else t.cpu().numpy()

    def _share3(self, share, obs=None):
        # the same shared vector for every trained agent (harlsustaindc_env.py:87 `repeat`): the 29-float layout the
        # kernel writes, or -- nonoverlapping_shared_obs_space False -- the concatenation of the trained agents' padded
        # observations (:83-85), which is a VIEW of the step's obs block (contiguous [N, 3, 26] -> [N, 78])
        if self.share_concat:
            o = self._sel_obs(obs)
            share = o.reshape(o.shape[0], self.share_dim)
        return share.unsqueeze(1).expand(-1, self.n_agents, -1)

    def _sel(self, x):
        # the trained agents' rows of a [N, 3, ...] array (all of them in the usual three-agent case)
        return x if self.n_agents == 3 else x[:, self._agent_idx]

    def _sel_obs(self, obs):
        # ... of the [N, 3, 26] observation block, cut to the widest trained agent's width
        o = self._sel(obs)
        return o if self.obs_width == L.OBS_PAD else o[:, :, :self.obs_width]

    def seed(self, seed: int):
        self.engine.set_seed(seed)

    # ------------------------------------------------------------------ ShareVecEnv API
    def reset(self):
        obs, share = self.engine.reset()
        self._need_reset = False
        return self._out(self._sel_obs(obs)), self._out(self._share3(share, obs)), (self._avail if self.return_torch else self._avail_np)

    def step_async(self, actions):
        t = self._torch
        if not isinstance(actions, t.Tensor):
            # host actions: through a pinned int32 staging buffer, asynchronously.  Two buffers alternate, and each carries
            # an event recorded behind its host->device copy: the buffer is rewritten only once that copy has run (with
            # device-resident outputs nothing else makes the host wait for the stream, so a caller that does not read
            # results could otherwise run several steps ahead and overwrite actions that have not been copied yet)
            if self._act_pin is None:
                self._act_pin = [t.empty((self.num_envs, self.n_agents), dtype=t.int32, pin_memory=True) for _ in range(2)]
                self._act_evt = [t.cuda.Event(), t.cuda.Event()]
                self._act_rec = [False, False]
            self._act_flip ^= 1
            k = self._act_flip
            pin = self._act_pin[k]
            if self._act_rec[k]:
                self._act_evt[k].synchronize()
            pin.numpy()[...] = np.asarray(actions).reshape(self.num_envs, self.n_agents)
            actions = pin.to(self.engine.device, non_blocking=True)
            self._act_evt[k].record(t.cuda.current_stream(self.engine.device))
            self._act_rec[k] = True
        a = actions.reshape(self.num_envs, self.n_agents)
        if a.dtype != t.int32 or a.device != self.engine.device:
            a = a.to(device=self.engine.device, dtype=t.int32)
        if self.n_agents != 3:      # the other slots are played on the device; their columns are never read
            full = t.ones((self.num_envs, 3), dtype=t.int32, device=self.engine.device)
            full[:, self._agent_idx] = a
            a = full
        self._actions = a.contiguous()

    def step_wait(self):
        if self._need_reset:
            raise RuntimeError("call reset() before step()")
        t = self._torch
        a = self._actions
        self._actions = None
        obs, share, rew, done, info = self.engine.step(a)
        if self.return_torch:
            # no host synchronisation on the device-resident path: which envs finished comes from the engine's host
            # mirror of the step counters (sdc_last_done)
            ld = self.engine.last_done()
            done_h = ld if ld is not None else self._no_done
        else:
            # NumPy outputs: ONE asynchronous copy of the step's outputs (one device allocation) into a pinned host buffer,
            # ONE synchronisation (two buffer sets alternate, so the arrays of a step stay valid until the step after next)
            hb = self._host_buffers()
            hb["flat"].copy_(self.engine.out_flat, non_blocking=True)     # the whole step in ONE device->host copy
            self._torch.cuda.current_stream(self.engine.device).synchronize()
            done_h = hb["done"].numpy().astype(bool)
        extra = {}
        if done_h.any():    # ONE host copy of the pre-reset observations; the per-env entries are built when read
            extra = _FinalObs(self.engine.final_obs.cpu().numpy(), done_h, self._agent_idx, self.n_agents, self.share_concat,
                              self.obs_width)
        # `infos` (lazy): reads the step's [N, 44] info block on first access -- from the pinned host copy made with the
        # other outputs (NumPy mode: valid for one more step), from the device otherwise.  An access after the block has
        # been overwritten RAISES instead of returning a later step's values; `snapshot_infos=True` makes every infos
        # object own a copy (the reference returns materialised dicts).
        self._gen += 1
        if self.return_torch:
            # device-resident path: a device clone only when asked for (snapshot_infos) or when an episode ended (the
            # runners read the final step's infos after the auto-reset); else a guarded view of the engine's buffer
            snap = self.snapshot_infos or bool(extra)
            infos = LazyInfos(info.clone() if snap else info, a, done_h, self._const, extra, None if snap else self, 0,
                              self.n_agents, self._info_keys)
        else:
            infos = LazyInfos(hb["info"], a, done_h, self._const, extra, self, 1, self.n_agents, self._info_keys)   # pinned double buffer: one more step
            if self.snapshot_infos:
                infos.rows()
        if self._logger_acc is not None:      # device-side logger sums: one small reduction per step, no read-back
            self._logger_acc.add_(info[:, self._logger_idx].sum(0, dtype=t.float64))
            self._logger_steps += 1
        k = self.n_agents
        if self.return_torch:
            # the engine's output tensors are persistent, so the shaped views are too (uint8 0/1 -> bool is a reinterpret)
            # (an agent SUBSET selects rows by index -- a copy, not a view: obs, rew and the concatenated shared observation
            # are then rebuilt every step)
            v = self._torch_views
            if v is None:
                share_is_view = (not self.share_concat) or k == 3
                v = (self._share3(share, obs) if share_is_view else None, done.view(t.bool).unsqueeze(1).expand(-1, k),
                     (obs, rew.unsqueeze(-1)) if k == 3 else None)
                self._torch_views = v
            o, r = v[2] if v[2] is not None else (self._sel_obs(obs), self._sel(rew).unsqueeze(-1))
            sh = v[0] if v[0] is not None else self._share3(share, obs)
            return o, sh, r, v[1], infos, self._avail
        if self.share_concat:
            sh = self._sel_obs(hb["obs"].numpy()).reshape(self.num_envs, self.share_dim)
        else:
            sh = hb["share"].numpy()
        share3 = np.broadcast_to(sh[:, None, :], (self.num_envs, k, sh.shape[1]))
        return (self._sel_obs(hb["obs"].numpy()), share3, self._sel(hb["rew"].numpy())[..., None],
                np.repeat(done_h[:, None], k, axis=1), infos, self._avail_np)

    def _launch_into(self, host):
        """One step enqueued, its outputs on their way into the caller's pinned host tensors (`host`: obs / share / rew / info
        / done slices of a larger block) -- nothing waited for.  For SustainDCMultiDeviceVecEnv: every device is given its
        work before any is waited on."""
        if self._need_reset:
            raise RuntimeError("call reset() before step()")
        a = self._actions
        self._actions = None
        e = self.engine
        e.step(a)
        host["obs"].copy_(e.obs, non_blocking=True)
        host["share"].copy_(e.share_obs, non_blocking=True)
        host["rew"].copy_(e.rew, non_blocking=True)
        host["info"].copy_(e.info, non_blocking=True)
        host["done"].copy_(e.done, non_blocking=True)
        if self._logger_acc is not None:
            self._logger_acc.add_(e.info[:, self._logger_idx].sum(0, dtype=self._torch.float64))
            self._logger_steps += 1
        return a

    def _host_buffers(self):
        t = self._torch
        if self._host is None:
            e = self.engine
            self._host = []
            for _ in range(2):
                flat = t.empty(e.out_flat.shape, dtype=t.uint8, pin_memory=True)
                o, sh, r, d, i = e.split_out_flat(flat)
                self._host.append({"flat": flat, "obs": o, "share": sh, "rew": r, "done": d, "info": i})
        self._host_flip ^= 1
        return self._host[self._host_flip]

    def accumulate_logger_sums(self, keys: Sequence[str] = LOGGER_KEYS, enable: bool = True):
        """Keep the SustainDC logger's per-step sums (harl/envs/sustaindc/sustaindc_logger.py:87-101: each key summed over
        the envs, every step) in a device-side accumulator: one small reduction per step on the stream, no host
        synchronisation; `read_logger_sums()` brings the totals over once per episode / log interval."""
        t = self._torch
        self._logger_keys = [k for k in keys]
        self._logger_idx = [L.INFO_IDX[k] for k in keys if k in L.INFO_IDX]
        self._logger_acc = t.zeros(len(self._logger_idx), dtype=t.float64, device=self.engine.device) if enable else None
        self._logger_steps = 0

    def read_logger_sums(self, reset: bool = True):
        """-> ({key: sum over envs and steps since the last read}, steps accumulated).  ONE device->host copy."""
        if self._logger_acc is None:
            raise RuntimeError("call accumulate_logger_sums() first")
        vals = self._logger_acc.cpu().numpy()
        out = {k: 0.0 for k in self._logger_keys}       # keys the env does not track (constant 0 in the reference too)
        out.update({k: float(v) for k, v in zip([k for k in self._logger_keys if k in L.INFO_IDX], vals)})
        n = self._logger_steps
        if reset:
            self._logger_acc.zero_()
            self._logger_steps = 0
        return out, n

    def info_sums(self, keys: Sequence[str] = LOGGER_KEYS):
        """Sum over envs of the given info columns for the last step, computed on the device."""
        idx = [L.INFO_IDX[k] for k in keys if k in L.INFO_IDX]
        s = self.engine.info[:, idx].sum(0).cpu().numpy()
        out = {k: 0.0 for k in keys}
        out.update({k: float(v) for k, v in zip([k for k in keys if k in L.INFO_IDX], s)})
        return out

    def close_extras(self):
        self.engine.close()
