"""SdcEngine: N SustainDC environment instances resident on one MI355X, driven through the C-ABI.

PyTorch is plumbing here: it owns the obs / action / reward / info device buffers and the HIP stream;
the step runs in the hand-written kernels of csrc/ (sdc_dynamics_kernel = one env-step of all N envs,
sdc_reset_kernel at episode boundaries).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib as L

_STATE_DTYPES = {
    "cursor": (np.int32, 1), "t_rel": (np.int32, 1), "day": (np.int32, 1), "hourq": (np.int32, 1),
    "q_popped": (np.int32, 1), "q_cum": (np.int32, 1), "q_cumT": (np.uint32, 1), "q_head": (np.int32, 1),
    "q_cum_hm1": (np.int32, 1), "q_cumT_hm1": (np.uint32, 1),
    "last_delta": (np.int32, 1), "consecutive": (np.int32, 1), "scale": (np.int32, 1),
    "hist_len": (np.int32, 1), "hist_pos": (np.int32, 1), "episode": (np.int32, 1), "fault": (np.uint32, 1),
    "loc_id": (np.int32, 1), "cfg_id": (np.int32, 1), "day_lo": (np.int32, 1), "day_hi": (np.int32, 1),
    "hist_n": (np.int32, 1), "order_stat_sticky": (np.uint32, 1),
    "stpt": (np.float64, 1), "bat_load": (np.float64, 1), "ci_min": (np.float64, 1), "ci_den": (np.float64, 1),
    "t_min": (np.float64, 1), "t_den": (np.float64, 1), "hist_ref": (np.float64, 1),
}
# a full checkpoint: the raw records + every array the kernels own
# "hist" first: injecting the ring drops the order-statistic trackers, which "header" then restores
_CHECKPOINT = ["hist", "record", "header", "qwin", "t_win", "wb_win", "qtab"]


def dc_params_struct(p: dict) -> L.SdcDcParams:
    """dict (see dc_config.size_datacenter) -> C struct."""
    s = L.SdcDcParams()
    R = len(p["rack_n"])
    if not 1 <= R <= L.MAX_RACKS:
        raise ValueError(f"n_racks must be in [1, {L.MAX_RACKS}], got {R}")
    s.n_racks = R
    for name in ("rack_n", "rack_full", "rack_idle", "rack_supply", "rack_return"):
        dst = getattr(s, name)
        src = p[name]
        if len(src) != R:
            raise ValueError(f"{name} has {len(src)} entries, expected {R}")
        for i in range(R):
            dst[i] = float(src[i])
    for name in ("m_cpu", "c_cpu", "rs_cpu", "m_fan", "c_fan", "rs_fan", "itfan_ref_p", "itfan_ref_v_ratio",
                 "it_fan_full_load_v", "c_air", "rho_air", "crac_supply_pu", "ct_fan_ref_p", "ctafr", "min_temp",
                 "max_temp"):
        setattr(s, name, float(p[name]))
    s.init_setpoint = float(p.get("init_setpoint", 18.0))
    s.bat_capacity_mwh = float(p["bat_capacity"])
    return s


_ACTOR_SD_KEYS = {   # the reference's StochasticPolicy state_dict -> sdc_actor_params fields
    "ln0_gamma": "base.feature_norm.weight", "ln0_beta": "base.feature_norm.bias",
    "w1": "base.mlp.fc.0.weight", "b1": "base.mlp.fc.0.bias", "ln1_gamma": "base.mlp.fc.2.weight", "ln1_beta": "base.mlp.fc.2.bias",
    "w2": "base.mlp.fc.3.weight", "b2": "base.mlp.fc.3.bias", "ln2_gamma": "base.mlp.fc.5.weight", "ln2_beta": "base.mlp.fc.5.bias",
    "w3": "act.action_out.linear.weight", "b3": "act.action_out.linear.bias",
}
_ACTOR_SHAPES = {"ln0_gamma": (26,), "ln0_beta": (26,), "w1": (64, 26), "b1": (64,), "ln1_gamma": (64,), "ln1_beta": (64,),
                 "w2": (64, 64), "b2": (64,), "ln2_gamma": (64,), "ln2_beta": (64,), "w3": (3, 64), "b3": (3,)}


def actor_params(params) -> L.SdcActorParams:
    """dict (sdc_actor_params field names, or the reference's StochasticPolicy state_dict keys) -> C struct."""
    p = L.SdcActorParams()
    feature_norm = True
    for field, shape in _ACTOR_SHAPES.items():
        v = params.get(field, params.get(_ACTOR_SD_KEYS[field]))
        if v is None:
            if field in ("ln0_gamma", "ln0_beta"):      # use_feature_normalization False: no feature_norm in the state_dict
                feature_norm = False
                continue
            raise KeyError(f"actor parameters: neither {field!r} nor {_ACTOR_SD_KEYS[field]!r} given")
        a = np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)
        if a.shape != shape:
            raise ValueError(f"actor parameter {field}: shape {a.shape}, expected {shape} (hidden_sizes [64, 64], 26 inputs, 3 actions)")
        C.memmove(getattr(p, field), a.ctypes.data, a.nbytes)
    p.use_feature_normalization = 1 if params.get("use_feature_normalization", feature_norm) else 0
    act = params.get("activation", "tanh")
    if act not in ("tanh", "relu", 0, 1):
        raise NotImplementedError(f"actor activation {act!r}: the kernel runs tanh (happo.yaml) and relu")
    p.activation = {"tanh": 0, "relu": 1}.get(act, act)
    return p


class SdcEngine:
    def __init__(self, n_envs: int, episode_steps: int = 672, device: int = 0, n_locations: int = 1,
                 n_dc_configs: int = 1, auto_reset: bool = True, seed: int = 0, hist_cap: int = 10000,
                 queue_max_len: int = 1000, weather_noise_std: float = 0.75, weather_noise_weight: float = 0.02,
                 max_roll_days: int = 14, debug_flags: int = 0, reward_method=(0, 0, 0), env_index_base: int = 0,
                 policy=(0, 0, 0), trim_and_respond_limit: float = 27.0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("SdcEngine needs an MI355X visible to PyTorch-ROCm (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.lib = L.load()
        self.torch = torch
        self.n_envs = int(n_envs)
        self.episode_steps = int(episode_steps)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        cfg = L.SdcConfig(n_envs=self.n_envs, device=self.device_index, episode_steps=self.episode_steps,
                          hist_cap=hist_cap, queue_max_len=queue_max_len, n_locations=n_locations,
                          n_dc_configs=n_dc_configs, auto_reset=1 if auto_reset else 0, seed=seed,
                          weather_noise_std=weather_noise_std, weather_noise_weight=weather_noise_weight,
                          max_roll_days=max_roll_days, debug_flags=debug_flags,
                          reward_method=(C.c_int32 * 3)(*[int(m) for m in reward_method]),
                          env_index_base=int(env_index_base), policy=(C.c_int32 * 3)(*[int(x) for x in policy]),
                          trim_and_respond_limit=float(trim_and_respond_limit))
        self.policy = tuple(int(x) for x in policy)
        self._h = C.c_void_p()
        self._pinned_stream = None
        self._pinned_stream_obj = None
        self._out_ptrs = None
        self._done_buf = None
        with torch.cuda.device(self.device):
            torch.cuda.init()
            L.check(self.lib.sdc_create(C.byref(cfg), C.byref(self._h)))
        self.lw = self.lib.sdc_weather_window_len(self._h)
        self.hist_stride = self.lib.sdc_hist_stride(self._h)
        self.queue_stride = self.lib.sdc_queue_stride(self._h)
        N = self.n_envs
        kw = dict(device=self.device)
        # the step's outputs are views of ONE device allocation (obs | share_obs | rew | info as floats, then done as
        # bytes), so that a host-side consumer can fetch a whole step with a single device->host copy (`out_flat`)
        n_f = N * (L.N_AGENTS * L.OBS_PAD + L.SHARE_OBS_DIM + L.N_AGENTS + L.INFO_DIM)
        self.out_flat = torch.zeros(n_f * 4 + N, dtype=torch.uint8, **kw)
        fl = self.out_flat[:n_f * 4].view(torch.float32)
        o = 0
        self.obs = fl[o:o + N * L.N_AGENTS * L.OBS_PAD].view(N, L.N_AGENTS, L.OBS_PAD); o += N * L.N_AGENTS * L.OBS_PAD
        self.share_obs = fl[o:o + N * L.SHARE_OBS_DIM].view(N, L.SHARE_OBS_DIM); o += N * L.SHARE_OBS_DIM
        self.rew = fl[o:o + N * L.N_AGENTS].view(N, L.N_AGENTS); o += N * L.N_AGENTS
        self.info = fl[o:o + N * L.INFO_DIM].view(N, L.INFO_DIM); o += N * L.INFO_DIM
        self.done = self.out_flat[n_f * 4:]
        self.final_obs = torch.zeros((N, L.N_AGENTS, L.OBS_PAD), dtype=torch.float32, **kw)

    # ------------------------------------------------------------------ setup
    def set_tables(self, loc_id: int, W, Cc, T, WB):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (W, Cc, T, WB)]
        for a in arrs:
            if a.shape != (L.TABLE_LEN,):
                raise ValueError(f"trace tables must have shape ({L.TABLE_LEN},), got {a.shape}")
        dp = C.POINTER(C.c_double)
        L.check(self.lib.sdc_set_tables(self._h, int(loc_id), *[a.ctypes.data_as(dp) for a in arrs], L.TABLE_LEN))

    def set_dc_params(self, cfg_id: int, params: dict):
        s = dc_params_struct(params)
        L.check(self.lib.sdc_set_dc_params(self._h, int(cfg_id), C.byref(s)))

    def assign(self, loc_id, cfg_id, day_lo, day_hi):
        N = self.n_envs
        arrs = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.int32), (N,))) for a in
                (loc_id, cfg_id, day_lo, day_hi)]
        ip = C.POINTER(C.c_int32)
        L.check(self.lib.sdc_assign_envs(self._h, *[a.ctypes.data_as(ip) for a in arrs]))

    def set_seed(self, seed: int):
        L.check(self.lib.sdc_set_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    # ------------------------------------------------------------------ run
    def _stream(self):
        # launches go to torch's current stream, or to the stream pinned with use_stream() (an engine per env group,
        # each on its own stream, lets the groups' steps overlap: tools/two_streams.py)
        if self._pinned_stream is not None:
            return self._pinned_stream
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def use_stream(self, stream=None):
        """Pin this engine's launches to a torch.cuda.Stream (None: back to torch's current stream)."""
        self._pinned_stream = None if stream is None else C.c_void_p(stream.cuda_stream)
        self._pinned_stream_obj = stream

    def reset(self, mask: Optional[np.ndarray] = None, override: Optional[dict] = None):
        """SustainDC.reset for the masked envs (all if mask is None).  Returns (obs, share_obs) device tensors
        (views of the engine's buffers)."""
        N = self.n_envs
        mptr = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            if m.shape != (N,):
                raise ValueError("mask must have shape (n_envs,)")
            mptr = m.ctypes.data_as(C.POINTER(C.c_uint8))
        optr = None
        keep = None
        if override is not None and "noise" in override:
            # the reference's own draws (day, hour, roll, the year's coherent-noise array): the device does the rest
            dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
            day = np.ascontiguousarray(override["day"], dtype=np.int32)
            hour = np.ascontiguousarray(override["hour"], dtype=np.int32)
            roll = np.ascontiguousarray(override["roll_days"], dtype=np.int32)
            noise = np.ascontiguousarray(override["noise"], dtype=np.float64)
            if noise.shape != (N, L.TABLE_LEN) or any(a.shape != (N,) for a in (day, hour, roll)):
                raise ValueError(f"noise injection: noise ({N}, {L.TABLE_LEN}), day / hour / roll_days ({N},)")
            nul = C.POINTER(C.c_double)()
            o = L.SdcResetOverride(day.ctypes.data_as(ip), hour.ctypes.data_as(ip), nul, nul, nul, nul, nul, nul,
                                   noise.ctypes.data_as(dp), roll.ctypes.data_as(ip))
            keep = (day, hour, roll, noise, o)
            optr = C.byref(o)
        elif override is not None:
            dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
            day = np.ascontiguousarray(override["day"], dtype=np.int32)
            hour = np.ascontiguousarray(override["hour"], dtype=np.int32)
            sc = [np.ascontiguousarray(override[k], dtype=np.float64) for k in ("ci_min", "ci_max", "t_min", "t_max")]
            tw = np.ascontiguousarray(override["t_win"], dtype=np.float64)
            wb = np.ascontiguousarray(override["wb_win"], dtype=np.float64)
            for a in [day, hour] + sc:
                if a.shape != (N,):
                    raise ValueError("override scalars must have shape (n_envs,)")
            if tw.shape != (N, self.lw) or wb.shape != (N, self.lw):
                raise ValueError(f"override weather windows must have shape ({N}, {self.lw})")
            o = L.SdcResetOverride(day.ctypes.data_as(ip), hour.ctypes.data_as(ip), *[a.ctypes.data_as(dp) for a in sc],
                                   tw.ctypes.data_as(dp), wb.ctypes.data_as(dp), C.POINTER(C.c_double)(),
                                   C.POINTER(C.c_int32)())
            keep = (day, hour, sc, tw, wb, o)
            optr = C.byref(o)
        with self.torch.cuda.device(self.device):
            L.check(self.lib.sdc_reset(self._h, mptr, optr, C.c_void_p(self.obs.data_ptr()),
                                       C.c_void_p(self.share_obs.data_ptr()), self._stream()))
        del keep
        return self.obs, self.share_obs

    def step(self, actions, want_info: bool = True):
        """actions: int32 device tensor [N, 3] (ls, dc, bat).  Returns views of the engine's buffers:
        obs [N,3,26], share_obs [N,29], rew [N,3], done [N] (uint8), info [N,40]."""
        t = self.torch
        if actions is None:
            if any(p == 0 for p in self.policy):
                raise ValueError("actions=None needs a built-in policy on every agent slot (SdcEngine(policy=...))")
        elif not (isinstance(actions, t.Tensor) and actions.dtype == t.int32 and actions.is_cuda and
                  actions.is_contiguous() and tuple(actions.shape) == (self.n_envs, 3)):
            raise ValueError("actions must be a contiguous int32 CUDA tensor of shape (n_envs, 3)")
        p = self._out_ptrs
        if p is None:   # the output tensors live as long as the engine: take their addresses once
            p = self._out_ptrs = tuple(C.c_void_p(x.data_ptr()) for x in
                                       (self.obs, self.share_obs, self.rew, self.done, self.info, self.final_obs))
        args = (self._h, C.c_void_p(actions.data_ptr()) if actions is not None else None, p[0], p[1], p[2], p[3],
                p[4] if want_info else None, p[5],
                self._stream())
        if t.cuda.current_device() == self.device_index:   # the usual case (one process per GPU): no device switch
            rc = self.lib.sdc_step(*args)
        else:
            with t.cuda.device(self.device):
                rc = self.lib.sdc_step(*args)
        if rc != 0:
            L.check(rc)
        return self.obs, self.share_obs, self.rew, self.done, self.info

    def split_out_flat(self, flat):
        """Views (obs, share_obs, rew, done, info) over a host / device copy of `out_flat` (a uint8 tensor of the same size)."""
        t, N = self.torch, self.n_envs
        n_f = N * (L.N_AGENTS * L.OBS_PAD + L.SHARE_OBS_DIM + L.N_AGENTS + L.INFO_DIM)
        fl = flat[:n_f * 4].view(t.float32)
        a = N * L.N_AGENTS * L.OBS_PAD
        b = a + N * L.SHARE_OBS_DIM
        c = b + N * L.N_AGENTS
        return (fl[:a].view(N, L.N_AGENTS, L.OBS_PAD), fl[a:b].view(N, L.SHARE_OBS_DIM), fl[b:c].view(N, L.N_AGENTS),
                flat[n_f * 4:], fl[c:].view(N, L.INFO_DIM))

    def last_step_kernel(self) -> str:
        """Name of the step kernel the last `step()` launched (the host picks by batch size and configuration; all give the same
        results)."""
        return self.lib.sdc_last_step_kernel(self._h).decode()

    def last_done(self):
        """bool [N]: which envs finished in the last step() / rollout() -- from the host's mirror of the step counters, no
        device synchronisation.  None when no env finished."""
        if self._done_buf is None:
            self._done_buf = np.zeros(self.n_envs, dtype=np.uint8)
            self._done_ptr = self._done_buf.ctypes.data_as(C.POINTER(C.c_uint8))
        n = self.lib.sdc_last_done(self._h, self._done_ptr)
        if n < 0:
            L.check(n)
        return self._done_buf.astype(bool) if n > 0 else None

    def steps_to_episode_end(self) -> int:
        return int(self.lib.sdc_steps_to_episode_end(self._h))

    def rollout_policy(self, n_steps: int, actions=None, want_info: bool = True):
        """`rollout` for engines whose agent slots (some or all) are played by built-in policies (`policy=`): closed-loop
        episodes at rollout speed.  actions: None when every slot has a policy, else [K, N, 3] (slots with a policy
        ignore their column).  Returns (obs, share_obs, rew, done, info, actions_applied [K, N, 3] int32)."""
        return self.rollout(actions, want_info=want_info, n_steps=n_steps, want_actions=True)

    def rollout(self, actions, want_info: bool = True, n_steps: int = None, want_actions: bool = False):
        """K env-steps in one launch for an action sequence known up front (scripted / rule-based policies, open-loop
        evaluation).  actions: int32 device tensor [K, N, 3]; K must not run past the end of an episode
        (steps_to_episode_end()).  Returns fresh device tensors holding every step's outputs:
        obs [K,N,3,26], share_obs [K,N,29], rew [K,N,3], done [K,N] (uint8), info [K,N,44] (or None).
        Same results as K calls of step()."""
        t = self.torch
        if actions is None:
            if n_steps is None or any(p == 0 for p in self.policy):
                raise ValueError("actions=None needs n_steps and a built-in policy on every agent slot")
            K = int(n_steps)
        else:
            if not (isinstance(actions, t.Tensor) and actions.dtype == t.int32 and actions.is_cuda and
                    actions.is_contiguous() and actions.dim() == 3 and tuple(actions.shape[1:]) == (self.n_envs, 3)):
                raise ValueError("actions must be a contiguous int32 CUDA tensor of shape (K, n_envs, 3)")
            K = int(actions.shape[0])
            if n_steps is not None and int(n_steps) != K:
                raise ValueError("n_steps does not match the action sequence")
        N = self.n_envs
        kw = dict(device=self.device)
        obs = t.empty((K, N, L.N_AGENTS, L.OBS_PAD), dtype=t.float32, **kw)
        share = t.empty((K, N, L.SHARE_OBS_DIM), dtype=t.float32, **kw)
        rew = t.empty((K, N, L.N_AGENTS), dtype=t.float32, **kw)
        done = t.empty((K, N), dtype=t.uint8, **kw)
        info = t.empty((K, N, L.INFO_DIM), dtype=t.float32, **kw) if want_info else None
        aout = t.empty((K, N, 3), dtype=t.int32, **kw) if want_actions else None
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        with t.cuda.device(self.device):
            L.check(self.lib.sdc_rollout(self._h, K, p(actions), p(obs), p(share), p(rew), p(done), p(info),
                                         p(self.final_obs), p(aout), self._stream()))
        # the engine's single-step views follow the last step
        self.obs.copy_(obs[-1]); self.share_obs.copy_(share[-1]); self.rew.copy_(rew[-1]); self.done.copy_(done[-1])
        if info is not None:
            self.info.copy_(info[-1])
        if want_actions:
            return obs, share, rew, done, info, aout
        return obs, share, rew, done, info

    # ------------------------------------------------------------------ closed loop: the actors inside the kernel
    def set_actor(self, agent_slot: int, params):
        """One agent's actor network (slot 0 agent_ls, 1 agent_dc, 2 agent_bat) for rollout_actor().  `params`: a
        state_dict of the reference's StochasticPolicy (harl/models/policy_models/stochastic_policy.py: keys
        base.feature_norm.*, base.mlp.fc.{0,2,3,5}.*, act.action_out.linear.*; tensors or arrays) or a dict with the
        fields of sdc_actor_params; `activation`: "tanh" (happo.yaml) or "relu"."""
        L.check(self.lib.sdc_set_actor(self._h, int(agent_slot), C.byref(actor_params(params))))

    def rollout_actor(self, n_steps: int, sample: bool = False, want_logits: bool = False):
        """K env-steps in ONE launch with the three actors (set_actor) choosing every action inside the kernel from the
        step's own observations -- the closed loop observation -> actor -> action -> step without a launch per step.
        sample=False: the distributions' mode (the reference's deterministic=True), True: a draw.
        Returns (obs [K,N,3,26], share_obs [K,N,29], rew [K,N,3], done [K,N], info [K,N,44], actions [K,N,3] int32,
        logits [K,N,3,3] or None).  K must not run past the end of the episode (steps_to_episode_end())."""
        t = self.torch
        K, N = int(n_steps), self.n_envs
        kw = dict(device=self.device)
        obs = t.empty((K, N, L.N_AGENTS, L.OBS_PAD), dtype=t.float32, **kw)
        share = t.empty((K, N, L.SHARE_OBS_DIM), dtype=t.float32, **kw)
        rew = t.empty((K, N, L.N_AGENTS), dtype=t.float32, **kw)
        done = t.empty((K, N), dtype=t.uint8, **kw)
        info = t.empty((K, N, L.INFO_DIM), dtype=t.float32, **kw)
        acts = t.empty((K, N, 3), dtype=t.int32, **kw)
        logits = t.empty((K, N, 3, 3), dtype=t.float32, **kw) if want_logits else None
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
        with t.cuda.device(self.device):
            L.check(self.lib.sdc_rollout_actor(self._h, K, 1 if sample else 0, p(obs), p(share), p(rew), p(done), p(info),
                                               p(self.final_obs), p(acts), p(logits), self._stream()))
        self.obs.copy_(obs[-1]); self.share_obs.copy_(share[-1]); self.rew.copy_(rew[-1]); self.done.copy_(done[-1])
        self.info.copy_(info[-1])
        return obs, share, rew, done, info, acts, logits

    # ------------------------------------------------------------------ state access (parity injection / checkpoint)
    def _state_array(self, name):
        N = self.n_envs
        if name in _STATE_DTYPES:
            dt, k = _STATE_DTYPES[name]
            return np.zeros((N,) if k == 1 else (N, k), dtype=dt)
        if name == "hist":
            return np.zeros((N, self.hist_stride), dtype=np.float32)
        if name in ("t_win", "wb_win"):
            return np.zeros((N, self.lw), dtype=np.float64)
        if name == "qtab":
            return np.zeros((N, self.queue_stride, 2), dtype=np.uint32)
        if name == "record":
            return np.zeros((N, 64), dtype=np.uint32)
        if name == "ep_return":
            return np.zeros((N, 3), dtype=np.float64)
        if name == "header":
            return np.zeros((N, L.HDR_DWORDS), dtype=np.uint32)
        if name == "qwin":
            return np.zeros((N, L.QWIN, 4), dtype=np.uint32)
        raise KeyError(name)

    def get_state(self, name: str) -> np.ndarray:
        a = self._state_array(name)
        L.check(self.lib.sdc_get_state(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def set_state(self, name: str, value):
        a = self._state_array(name)
        a[...] = value
        L.check(self.lib.sdc_set_state(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def state_dict(self) -> dict:
        """Full env checkpoint (the reference never checkpoints env state)."""
        return {n: self.get_state(n) for n in _CHECKPOINT}

    def load_state_dict(self, sd: dict):
        for n, v in sd.items():
            self.set_state(n, v)

    def profile(self, every: int = 1):
        """Per-kernel HIP-event timing on the launch stream (measurement only): every k-th step, 0 = off."""
        L.check(self.lib.sdc_profile_enable(self._h, int(every)))

    def profile_read(self, reset: bool = True) -> dict:
        out = (C.c_double * 5)()
        L.check(self.lib.sdc_profile_read(self._h, out, 1 if reset else 0))
        return {"dynamics_ms": out[0], "reward_ms": out[1], "reset_ms": out[2], "steps": int(out[3]),
                "resets": int(out[4])}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.sdc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
