#!/usr/bin/env python3
"""Generate golden input/output vectors from the *imported* Python reference.

Runs ONLY in the build container (needs /root/reference); the resulting small
`.npz` fixtures are committed under tests/golden/ and are what travels to the GPU
box.  Nothing from the reference's sources is copied: the fixtures hold inputs
(trace-table windows as the reference constructed them, rack table, sized HVAC
constants, initial state, the action sequence) and expected outputs (53 obs floats,
3 rewards, done flag and ~35 info scalars per step).

Determinism fixes applied to the reference:
  * rack order frozen: `utils.dc_config_reader.as_completed` -> submission order
    (the reference's ThreadPool `as_completed` order is non-deterministic,
    /root/reference/utils/dc_config_reader.py:100-105);
  * `reward_creator.energy_history.clear()` per fixture (module-global deque,
    /root/reference/utils/reward_creator.py:5);
  * `random.seed(s); np.random.seed(s)` per fixture (reset uses both:
    /root/reference/sustaindc_env.py:454-455, utils/managers.py:45,601).

Usage:  python tests/golden/gen_golden.py [--only NAME]
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SDC_REFERENCE", "/root/reference")
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))

INFO_KEYS = [
    "ls_original_workload", "ls_shifted_workload", "ls_tasks_in_queue", "ls_norm_tasks_in_queue",
    "ls_tasks_dropped", "ls_tasks_processed", "ls_oldest_task_age", "ls_average_task_age",
    "ls_overdue_penalty", "ls_computed_tasks", "ls_current_hour",
    "dc_ITE_total_power_kW", "dc_CT_total_power_kW", "dc_Compressor_total_power_kW",
    "dc_HVAC_total_power_kW", "dc_total_power_kW", "dc_crac_setpoint_delta", "dc_crac_setpoint",
    "dc_cpu_workload_fraction", "dc_int_temperature", "dc_exterior_ambient_temp", "dc_water_usage",
    "bat_action", "bat_SOC", "bat_CO2_footprint", "bat_avg_CI",
    "bat_total_energy_without_battery_KWh", "bat_total_energy_with_battery_KWh",
    "norm_CI", "outside_temp", "day", "hour",
]


def _install_shims():
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, REF)
    # sustaindc_env.py:32 imports the Dash dashboard through harl/envs/__init__.py, which
    # pulls absl / dash_bootstrap_components (absent).  Pre-register a dummy module.
    for name in ("harl", "harl.envs", "harl.envs.sustaindc"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    dash = types.ModuleType("harl.envs.sustaindc.dashboard_v2")

    class Dashboard:  # pragma: no cover - never started
        def __init__(self, *a, **k):
            pass

        def start(self):
            pass

    dash.Dashboard = Dashboard
    sys.modules["harl.envs.sustaindc.dashboard_v2"] = dash
    import utils.dc_config_reader as dcr
    dcr.as_completed = lambda futures: list(futures)  # freeze rack order = JSON order


def _static_block(env):
    """As-constructed per-rack table and init-time constants."""
    dc = env.dc_env
    cfg = dc.DC_Config
    racks = dc.dc.racks_list
    out = {
        "rack_n": np.array([r.num_CPUs for r in racks], dtype=np.float64),
        "rack_full": np.array([r.full_load_pwr[0] for r in racks], dtype=np.float64),
        "rack_idle": np.array([r.idle_pwr[0] for r in racks], dtype=np.float64),
        "rack_supply": np.array(cfg.RACK_SUPPLY_APPROACH_TEMP_LIST, dtype=np.float64),
        "rack_return": np.array(cfg.RACK_RETURN_APPROACH_TEMP_LIST, dtype=np.float64),
        # every CPU in a rack is identical; record the scalars the rack model uses
        "m_cpu": racks[0].m_cpu[0], "c_cpu": racks[0].c_cpu[0], "rs_cpu": racks[0].ratio_shift_max_cpu[0],
        "m_fan": racks[0].m_itfan[0], "c_fan": racks[0].c_itfan[0], "rs_fan": racks[0].ratio_shift_max_itfan[0],
        "itfan_ref_p": cfg.ITFAN_REF_P, "itfan_ref_v_ratio": cfg.ITFAN_REF_V_RATIO,
        "it_fan_full_load_v": cfg.IT_FAN_FULL_LOAD_V,
        "c_air": cfg.C_AIR, "rho_air": cfg.RHO_AIR,
        "crac_supply_pu": cfg.CRAC_SUPPLY_AIR_FLOW_RATE_pu,
        "ct_fan_ref_p": cfg.CT_FAN_REF_P,            # sized (make_envs_pyenv.py:159-161)
        "ctafr": cfg.CT_REFRENCE_AIR_FLOW_RATE,      # sized
        "min_temp": dc.min_temp, "max_temp": dc.max_temp,
        "power_lb_kW": dc.power_lb_kW, "power_ub_kW": dc.power_ub_kW,
        "bat_capacity": env.bat_env.battery.capacity,
        "bat_dcload_min": env.bat_env.dcload_min, "bat_dcload_max": env.bat_env.dcload_max,
        "range_zone_air": np.array(dc.ranges["Zone Air Temperature(West Zone)"], dtype=np.float64),
        "range_hvac": np.array(dc.ranges["Facility Total HVAC Electricity Demand Rate(Whole Building)"], dtype=np.float64),
        "range_total": np.array(dc.ranges["Facility Total Electricity Demand Rate(Whole Building)"], dtype=np.float64),
        "range_it": np.array(dc.ranges["Facility Total Building Electricity Demand Rate(Whole Building)"], dtype=np.float64),
        "init_day": env.init_day,
        "queue_max_len": env.ls_env.queue_max_len,
    }
    return {k: np.asarray(v) for k, v in out.items()}


def _policy(name, steps, rng):
    a = rng.integers(0, 3, size=(steps, 3))
    if name == "random":
        pass
    elif name == "defer":
        a[:, 0] = 0
    elif name == "process":
        a[:, 0] = 2
    elif name == "defer_drain":
        a[:, 0] = 0
        a[min(300, steps // 2):, 0] = 2
    elif name == "stpt_up":
        a[:, 1] = 2
    elif name == "stpt_down":
        a[:, 1] = 0
    elif name == "stpt_saw":
        blk = (np.arange(steps) // 9) % 2
        a[:, 1] = np.where(blk == 0, 2, 0)
    elif name == "bat_cycle":
        blk = (np.arange(steps) // 50) % 2
        a[:, 2] = np.where(blk == 0, 0, 1)
    elif name == "idle":
        a[:, 0] = 1
        a[:, 1] = 1
        a[:, 2] = 2
    else:
        raise ValueError(name)
    return a.astype(np.int32)


def _flat_obs(obs):
    return np.concatenate([obs["agent_ls"], obs["agent_dc"], obs["agent_bat"]]).astype(np.float32)


def _episode_inputs(env, steps):
    """Trace-table windows exactly as the reference's managers hold them after reset()."""
    c0 = env.ci_m.time_step
    lo = max(0, c0 - 16)
    hi = c0 + steps + 18
    ci = env.ci_m
    we = env.weather_m
    wl = env.workload_m
    assert hi <= len(ci.carbon_smooth), "episode runs past the year table (reference would raise IndexError)"
    return {
        "cursor0": c0, "win_lo": lo,
        "W": np.array(wl.cpu_smooth[lo:hi], dtype=np.float64),
        "C": np.array(ci.carbon_smooth[lo:hi], dtype=np.float64),
        "NC": np.array(ci.norm_carbon[lo:hi], dtype=np.float64),
        "T": np.array(we.temperature_data[lo:hi], dtype=np.float64),
        "WB": np.array(we.wet_bulb_data[lo:hi], dtype=np.float64),
        "NT": np.array(we.norm_temp_data[lo:hi], dtype=np.float64),
        "ci_min30": np.min(ci.carbon_smooth[c0:c0 + 2880]), "ci_max30": np.max(ci.carbon_smooth[c0:c0 + 2880]),
        "t_min30": np.min(we.temperature_data[c0:c0 + 2880]), "t_max30": np.max(we.temperature_data[c0:c0 + 2880]),
        "init_day": env.t_m.day, "init_hour": env.t_m.hour,
    }


def run_fixture(spec):
    from utils import reward_creator
    from sustaindc_env import SustainDC

    seed = spec["seed"]
    reward_creator.energy_history.clear()
    random.seed(seed)
    np.random.seed(seed)
    cfg = {
        "location": spec["location"], "month": spec["month"],
        "days_per_episode": spec.get("days", 7),
        "datacenter_capacity_mw": spec.get("capacity_mw", 1),
        "dc_config_file": spec.get("dc_config_file", "dc_config.json"),
        "agents": ["agent_ls", "agent_dc", "agent_bat"],
    }
    if "rewards" in spec:   # alternate reward functions (utils/reward_creator.py:322-334)
        cfg["ls_reward"], cfg["dc_reward"], cfg["bat_reward"] = spec["rewards"]
    env = SustainDC(cfg)
    if "force_day_range" in spec:
        env.ranges_day = list(spec["force_day_range"])   # public attribute (sustaindc_env.py:198)
    steps = cfg["days_per_episode"] * 96
    n_ep = spec.get("episodes", 1)
    rng = np.random.default_rng(seed)
    out = {f"static_{k}": v for k, v in _static_block(env).items()}
    out["meta_location"] = np.array(spec["location"])
    out["meta_month"] = np.array(spec["month"])
    out["meta_steps"] = np.array(steps)
    out["meta_episodes"] = np.array(n_ep)
    out["meta_seed"] = np.array(seed)
    out["meta_policy"] = np.array(spec["policy"])
    out["meta_capacity_mw"] = np.array(float(cfg["datacenter_capacity_mw"]))
    out["meta_dc_config"] = np.array(os.path.basename(cfg["dc_config_file"]))
    out["meta_days"] = np.array(cfg["days_per_episode"])
    out["meta_info_keys"] = np.array(INFO_KEYS)
    if "rewards" in spec:
        out["meta_reward_names"] = np.array(list(spec["rewards"]))
        out["meta_reward_method"] = np.array([REWARD_CODES[i][m] for i, m in enumerate(spec["rewards"])], dtype=np.int32)
    out["init_stpt"] = np.array(env.dc_env.raw_curr_stpt, dtype=np.float64)

    reset_obs = _flat_obs(env.reset())
    for ep in range(n_ep):
        if spec.get("need_early_cursor") and env.ci_m.time_step >= 15:
            raise RuntimeError("seed does not give cursor < 15; pick another")
        inputs = _episode_inputs(env, steps)
        acts = _policy(spec["policy"], steps, rng)
        obs = np.zeros((steps, 53), dtype=np.float32)
        rew = np.zeros((steps, 3), dtype=np.float64)
        done = np.zeros(steps, dtype=np.uint8)
        info = np.zeros((steps, len(INFO_KEYS)), dtype=np.float64)
        hist = np.zeros((steps, 5), dtype=np.float64)
        hist_len0 = len(reward_creator.energy_history)
        stpt0 = env.dc_env.raw_curr_stpt
        for t in range(steps):
            a = acts[t]
            o, r, term, trunc, inf = env.step({"agent_ls": int(a[0]), "agent_dc": int(a[1]), "agent_bat": int(a[2])})
            obs[t] = _flat_obs(o)
            rew[t] = [r["agent_ls"], r["agent_dc"], r["agent_bat"]]
            done[t] = 1 if (trunc["__all__"] or term["__all__"]) else 0
            common = inf["agent_ls"]
            info[t] = [float(common[k]) for k in INFO_KEYS]
            hist[t] = np.asarray(common["ls_task_age_histogram"], dtype=np.float64)
        assert done[-1] == 1 and done[:-1].sum() == 0
        p = f"ep{ep}_"
        for k, v in inputs.items():
            out[p + k] = np.asarray(v)
        out[p + "reset_obs"] = reset_obs
        out[p + "actions"] = acts
        out[p + "obs"] = obs
        out[p + "rew"] = rew
        out[p + "done"] = done
        out[p + "info"] = info
        out[p + "age_hist"] = hist
        out[p + "hist_len0"] = np.array(hist_len0)
        out[p + "stpt0"] = np.array(stpt0, dtype=np.float64)
        # HARL auto-reset (harl/envs/env_wrappers.py:176-190): reset inside the same step call
        reset_obs = _flat_obs(env.reset())
    out["final_energy_history_tail"] = np.array(list(reward_creator.energy_history)[-64:], dtype=np.float64)
    return out


VARIANT_DIR = os.path.join(REPO, "dc_rl_amd", "configs")

# reward-method codes of include/sustaindc_hip.h (sdc_config.reward_method), per agent slot
_ALT = {"custom_agent_reward": 2, "tou_reward": 3, "energy_efficiency_reward": 4, "energy_PUE_reward": 5,
        "water_usage_efficiency_reward": 6}
REWARD_CODES = [
    {"default_ls_reward": 0, "default_dc_reward": 1, "default_bat_reward": 1, **_ALT},
    {"default_dc_reward": 0, "default_bat_reward": 1, **_ALT},
    {"default_bat_reward": 0, "default_dc_reward": 1, **_ALT},
]

SPECS = {
    # BASELINE.json config 1: NY, Alibaba trace, month 6, 7 days, seed 0, uniform-random actions
    "ny_m6_random": dict(location="ny", month=6, seed=0, policy="random"),
    "ny_m0_defer": dict(location="ny", month=0, seed=1, policy="defer"),
    "ca_m3_defer_drain": dict(location="ca", month=3, seed=2, policy="defer_drain"),
    "az_m7_stpt_saw": dict(location="az", month=7, seed=3, policy="stpt_saw"),
    "wa_m11_bat_cycle": dict(location="wa", month=11, seed=4, policy="bat_cycle"),
    "tx_m9_stpt_down_2mw": dict(location="tx", month=9, seed=5, policy="stpt_down", capacity_mw=2),
    "il_m1_process": dict(location="il", month=1, seed=6, policy="process"),
    # 0.5 MW: MAX_W_PER_RACK = 25 kW caps the CPUs per rack (datacenter.py:67-74) -> ragged n_r
    "ga_m4_idle_halfmw": dict(location="ga", month=4, seed=11, policy="idle", capacity_mw=0.5),
    # cursor < 16 edge (get_n_past_ci returns an empty slice, managers.py:482-483)
    "ny_m0_early_cursor": dict(location="ny", month=0, seed=None, policy="random", force_day_range=(0, 0),
                               need_early_cursor=True),
    # history deque crosses 10 000, set-point carried across resets (16 x 672 = 10 752 steps)
    "ny_m6_multi16": dict(location="ny", month=6, seed=7, policy="random", episodes=16),
    # HARL YAML shape: 30-day episode, location ca
    "ca_m6_30day": dict(location="ca", month=6, seed=8, policy="defer_drain", days=30),
    # heterogeneous rack-count variants (our own JSONs; BASELINE config 4)
    "ny_m5_r16": dict(location="ny", month=5, seed=9, policy="random",
                      dc_config_file=os.path.join(VARIANT_DIR, "dc_config_r16.json")),
    "ny_m5_r25": dict(location="ny", month=5, seed=10, policy="stpt_up",
                      dc_config_file=os.path.join(VARIANT_DIR, "dc_config_r25.json")),
    # alternate reward functions (REWARD_METHOD_MAP); the second one never calls default_ls_reward, so the energy
    # history stays empty and the footprint reward of agent_bat is identically 0
    "ny_m8_alt_rewards": dict(location="ny", month=8, seed=12, policy="random",
                              rewards=("default_ls_reward", "energy_PUE_reward", "water_usage_efficiency_reward")),
    "az_m2_alt_rewards_nohist": dict(location="az", month=2, seed=13, policy="random", episodes=2,
                                     rewards=("energy_efficiency_reward", "custom_agent_reward", "default_bat_reward")),
}


# ---------------------------------------------------------------------------------------------------------------------
# weather fixture: pre-noise tables + reference resets

WB_NOTE = ("*_WB / WB_win arrays came through the generator's psychrolib shim (tests/golden/_shims/psychrolib.py), i.e. "
           "through dc_rl_amd/psychro.py: they pin the interpolation / roll / noise plumbing, NOT PsychroLib's wet-bulb routine")


def gen_weather_resets():
    """`Weather_Manager` (utils/managers.py:488-628) for three locations: the pre-noise 15-minute tables it keeps
    (`original_temp_data` / `original_wb_data`, :560-561) and, for a few seeded resets, the coherent noise, the roll and
    the resulting episode windows / 30-day normalisation bounds.  The noise itself is not stored (280 KB per reset): it
    is `np.random.seed(s); np.cumsum(0.02 * np.random.normal(0, 1, 35040))` rescaled (managers.py:45-47) -- NumPy's
    legacy MT19937 stream is frozen -- and the fixture keeps every 32nd sample so a test can confirm that it regenerated
    the same array."""
    from utils.managers import Weather_Manager
    from utils.utils_cf import obtain_paths
    out = {"meta_locations": np.array(["ny", "az", "wa"])}
    cases = [("ny", 101, 181, 13, 0), ("az", 102, 10, 0, 0), ("wa", 103, 300, 23, 0), ("ny", 104, 333, 5, 5)]
    mgr = {}
    for loc in ("ny", "az", "wa"):
        _, wea = obtain_paths(loc)
        for tz in sorted({c[4] for c in cases if c[0] == loc}):
            wm = Weather_Manager(location=wea, timezone_shift=tz)
            mgr[(loc, tz)] = wm
            if tz == 0:
                out[f"{loc}_T"] = np.asarray(wm.original_temp_data, dtype=np.float64)
                out[f"{loc}_WB"] = np.asarray(wm.original_wb_data, dtype=np.float64)
    W = 2880 + 18 + 96    # episode window of a 30-day episode (+18) and a day beyond
    for k, (loc, seed, day, hour, tz) in enumerate(cases):
        wm = mgr[(loc, tz)]
        np.random.seed(seed)
        wm.reset(init_day=day, init_hour=hour)
        c0 = wm.time_step
        # replay the same draws to capture what reset() consumed (same call sequence: generate(), then randint)
        np.random.seed(seed)
        noise = wm.coherent_noise.generate(len(wm.original_temp_data))
        roll = int(np.random.randint(0, 14))
        t_full = np.clip(np.roll(wm.original_temp_data + noise, roll * 96), 0, 45)
        assert np.array_equal(t_full, wm.temperature_data), "replayed draws do not reproduce the reset"
        pre = f"case{k}_"
        out[pre + "loc"] = np.array(loc)
        out[pre + "seed"] = np.array(seed)
        out[pre + "day"] = np.array(day)
        out[pre + "hour"] = np.array(hour)
        out[pre + "tz"] = np.array(tz)
        out[pre + "cursor0"] = np.array(c0)
        out[pre + "roll_days"] = np.array(roll)
        out[pre + "noise_sub32"] = np.asarray(noise[::32], dtype=np.float64)
        out[pre + "noise_std_target"] = np.array(float(np.std(noise)))
        out[pre + "T_win"] = np.asarray(wm.temperature_data[c0:c0 + W], dtype=np.float64)
        out[pre + "WB_win"] = np.asarray(wm.wet_bulb_data[c0:c0 + W], dtype=np.float64)
        out[pre + "NT_win"] = np.asarray(wm.norm_temp_data[c0:c0 + W], dtype=np.float64)
        out[pre + "t_min30"] = np.array(np.min(wm.temperature_data[c0:c0 + 2880]))
        out[pre + "t_max30"] = np.array(np.max(wm.temperature_data[c0:c0 + 2880]))
        if tz != 0:
            out[pre + "T_orig_sub16"] = np.asarray(wm.original_temp_data[::16], dtype=np.float64)
    out["meta_cases"] = np.array(len(cases))
    out["meta_window"] = np.array(W)
    out["meta_wb_source"] = np.array(WB_NOTE)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# HARL layer fixture: the reference's own HARLSustainDCEnv + ShareDummyVecEnv

def _load_by_path(name, relpath):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def gen_harl_layer(n_envs=4, days=1, n_steps=200, seed=21, nonoverlapping=True):
    """N reference environments behind the reference's HARL adaptation layer (harlsustaindc_env.py:10-131 over
    sustaindc_ptzoo.py, with pad_observations_v0) and its in-process vector wrapper (env_wrappers.py:301-350,
    auto-reset at :321-343), built by the rule of harl/utils/envs_tools.py:49-71 (month = rank, seed + rank * 1000).
    Records what the runner sees -- obs [N,3,26], share_obs [N,3,29], rews [N,3,1], dones [N,3], available actions,
    `original_obs` / `original_state` on the done step -- plus every episode's inputs for injection.

    The reference runs one OS process per env, so `reward_creator.energy_history` (a module global) is per env; with
    ShareDummyVecEnv the envs share one process, so each env gets its own deque here (swapped in around its calls).

    nonoverlapping=False records the layer's OTHER shared-observation option (harlsustaindc_env.py:53, :83-85: the
    concatenation of the three padded observations, 78 floats; share space Box(0, 1, (78,)) of sustaindc_ptzoo.py:32-44)."""
    import collections
    from utils import reward_creator
    _load_by_path("harl.envs.sustaindc.sustaindc_ptzoo", "harl/envs/sustaindc/sustaindc_ptzoo.py")
    hmod = _load_by_path("harl.envs.sustaindc.harlsustaindc_env", "harl/envs/sustaindc/harlsustaindc_env.py")
    wmod = _load_by_path("harl.envs.env_wrappers", "harl/envs/env_wrappers.py")
    steps = days * 96
    base_args = {"location": "ny", "days_per_episode": days, "datacenter_capacity_mw": 1, "dc_config_file": "dc_config.json",
                 "agents": ["agent_ls", "agent_dc", "agent_bat"], "partial_obs": True,
                 "nonoverlapping_shared_obs_space": bool(nonoverlapping)}

    class OwnHistory:
        """One env with its own energy history (= its own process in the reference's ShareSubprocVecEnv)."""

        def __init__(self, env):
            self.env = env
            self.hist = collections.deque(maxlen=10000)
            self.observation_space, self.share_observation_space = env.observation_space, env.share_observation_space
            self.action_space, self.n_agents = env.action_space, env.n_agents

        def step(self, a):
            reward_creator.energy_history = self.hist
            return self.env.step(a)

        def reset(self):
            reward_creator.energy_history = self.hist
            return self.env.reset()

        def close(self):
            pass

    random.seed(seed)
    np.random.seed(seed)

    def get_env_fn(rank):
        def init_env():
            args = dict(base_args)
            args["month"] = rank % 12 if rank < 12 else rank % 3 + 5
            env = hmod.HARLSustainDCEnv(args)
            env.seed(seed + rank * 1000)
            return OwnHistory(env)
        return init_env

    venv = wmod.ShareDummyVecEnv([get_env_fn(i) for i in range(n_envs)])
    inner = [e.env.env.unwrapped.env for e in venv.envs]     # the SustainDC objects
    rng = np.random.default_rng(seed)
    out = {f"static_{k}": v for k, v in _static_block(inner[0]).items()}
    out["meta_n_envs"] = np.array(n_envs)
    out["meta_steps"] = np.array(steps)
    out["meta_n_steps"] = np.array(n_steps)
    out["meta_months"] = np.array([e.month for e in inner])
    out["init_stpt"] = np.array(inner[0].dc_env.raw_curr_stpt, dtype=np.float64)
    obs, share, avail = venv.reset()
    out["reset_obs"], out["reset_share"], out["reset_avail"] = (np.asarray(x, dtype=np.float32) for x in (obs, share, avail))
    n_ep = np.zeros(n_envs, dtype=int)

    def record_inputs(i):
        for k, v in _episode_inputs(inner[i], steps).items():
            out[f"env{i}_ep{n_ep[i]}_{k}"] = np.asarray(v)
        n_ep[i] += 1

    for i in range(n_envs):
        record_inputs(i)
    acts = rng.integers(0, 3, size=(n_steps, n_envs, 3, 1))
    O = np.zeros((n_steps, n_envs, 3, 26), np.float32)
    sdim = int(venv.share_observation_space[0].shape[0])
    S = np.zeros((n_steps, n_envs, 3, sdim), np.float32)
    R = np.zeros((n_steps, n_envs, 3, 1), np.float64)
    D = np.zeros((n_steps, n_envs, 3), np.uint8)
    A = np.zeros((n_steps, n_envs, 3, 3), np.float32)
    OO = np.zeros((n_steps, n_envs, 3, 26), np.float32)
    OS = np.zeros((n_steps, n_envs, 3, sdim), np.float32)
    for t in range(n_steps):
        obs, share, rews, dones, infos, avail = venv.step(acts[t])
        O[t], S[t], R[t], D[t], A[t] = obs, share, rews, dones, avail
        for i in range(n_envs):
            if np.all(dones[i]):
                OO[t, i] = infos[i][0]["original_obs"]
                OS[t, i] = infos[i][0]["original_state"]
                assert np.array_equal(infos[i][0]["original_avail_actions"], np.ones((3, 3)))
                record_inputs(i)      # the wrapper has already reset the env: its managers hold the next episode
    out.update(actions=acts.astype(np.int32), obs=O, share_obs=S, rews=R, dones=D, avail=A, original_obs=OO, original_state=OS)
    out["meta_episodes"] = n_ep
    out["share_space_shape"] = np.array(venv.share_observation_space[0].shape)
    out["share_space_low_high"] = np.array([venv.share_observation_space[0].low.min(), venv.share_observation_space[0].high.max()])
    out["meta_nonoverlapping"] = np.array(bool(nonoverlapping))
    out["obs_space_shape"] = np.array(venv.observation_space[0].shape)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# rule-based baseline episode: the reference env driven by the reference's own controllers

def gen_rbc_episode(location="ny", month=7, seed=31, episodes=2, tr_limit=None):
    """SustainDC stepped closed-loop by the reference's rule-based agents: BaseLoadShiftingAgent (do nothing),
    trim_and_respond_ctrl on agent_dc (utils/trim_and_respond.py, fed the dc_int_temperature of the previous step's info,
    0.0 before the first step; its counter is never reset) and RBCBatteryAgent on agent_bat (utils/rbc_agents.py, fed
    [ci] + ci_future of `env.infos["__common__"]`, sustaindc_env.py:601-603).  An ordinary episode fixture whose
    `actions` are what the controllers chose."""
    from utils import reward_creator
    from utils.base_agents import BaseLoadShiftingAgent
    from utils.rbc_agents import RBCBatteryAgent
    from utils.trim_and_respond import trim_and_respond_ctrl
    from sustaindc_env import SustainDC
    reward_creator.energy_history.clear()
    random.seed(seed)
    np.random.seed(seed)
    env = SustainDC({"location": location, "month": month, "days_per_episode": 7, "datacenter_capacity_mw": 1,
                     "dc_config_file": "dc_config.json", "agents": ["agent_ls", "agent_dc", "agent_bat"]})
    steps = 7 * 96
    ls_agent, bat_agent = BaseLoadShiftingAgent(), RBCBatteryAgent()
    out = {f"static_{k}": v for k, v in _static_block(env).items()}
    reset_obs = _flat_obs(env.reset())
    if tr_limit is None:      # a limit both branches of the controller see: the median room temperature of a dry run
        probe = []
        for t in range(96):
            o, r, te, tr, inf = env.step({"agent_ls": 1, "agent_dc": 1, "agent_bat": 2})
            probe.append(float(inf["agent_ls"]["dc_int_temperature"]))
        tr_limit = float(np.round(np.median(probe), 1))
        reward_creator.energy_history.clear()
        random.seed(seed)
        np.random.seed(seed)
        env = SustainDC({"location": location, "month": month, "days_per_episode": 7, "datacenter_capacity_mw": 1,
                         "dc_config_file": "dc_config.json", "agents": ["agent_ls", "agent_dc", "agent_bat"]})
        reset_obs = _flat_obs(env.reset())
    dc_agent = trim_and_respond_ctrl(TandR_monitor_limit=tr_limit)
    out.update(meta_location=np.array(location), meta_month=np.array(month), meta_steps=np.array(steps),
               meta_episodes=np.array(episodes), meta_seed=np.array(seed), meta_policy=np.array("rbc"),
               meta_capacity_mw=np.array(1.0), meta_dc_config=np.array("dc_config.json"), meta_days=np.array(7),
               meta_info_keys=np.array(INFO_KEYS), meta_tr_limit=np.array(tr_limit),
               meta_device_policy=np.array([1, 3, 2], dtype=np.int32),
               init_stpt=np.array(env.dc_env.raw_curr_stpt, dtype=np.float64))
    last_room = 0.0
    for ep in range(episodes):
        inputs = _episode_inputs(env, steps)
        acts = np.zeros((steps, 3), np.int32)
        obs = np.zeros((steps, 53), dtype=np.float32)
        rew = np.zeros((steps, 3), dtype=np.float64)
        done = np.zeros(steps, dtype=np.uint8)
        info = np.zeros((steps, len(INFO_KEYS)), dtype=np.float64)
        hist = np.zeros((steps, 5), dtype=np.float64)
        hist_len0 = len(reward_creator.energy_history)
        stpt0 = env.dc_env.raw_curr_stpt
        for t in range(steps):
            common = env.infos["__common__"]
            ci_values = np.concatenate([[common["ci"]], np.asarray(common["ci_future"])])
            soc = env.bat_env.get_battery_soc() if hasattr(env.bat_env, "get_battery_soc") else 0.0
            a = [int(ls_agent.do_nothing_action()), int(dc_agent.action(last_room)), int(bat_agent.act(ci_values, soc))]
            acts[t] = a
            o, r, term, trunc, inf = env.step({"agent_ls": a[0], "agent_dc": a[1], "agent_bat": a[2]})
            obs[t] = _flat_obs(o)
            rew[t] = [r["agent_ls"], r["agent_dc"], r["agent_bat"]]
            done[t] = 1 if (trunc["__all__"] or term["__all__"]) else 0
            c = inf["agent_ls"]
            info[t] = [float(c[k]) for k in INFO_KEYS]
            hist[t] = np.asarray(c["ls_task_age_histogram"], dtype=np.float64)
            last_room = float(c["dc_int_temperature"])
        p = f"ep{ep}_"
        for k, v in inputs.items():
            out[p + k] = np.asarray(v)
        out.update({p + "reset_obs": reset_obs, p + "actions": acts, p + "obs": obs, p + "rew": rew, p + "done": done,
                    p + "info": info, p + "age_hist": hist, p + "hist_len0": np.array(hist_len0),
                    p + "stpt0": np.array(stpt0, dtype=np.float64)})
        reset_obs = _flat_obs(env.reset())
    return out


def gen_tou_prices():
    """utils/reward_creator.py:154-198 called DIRECTLY on whole-hour parameters: the 24 prices of its table (energy 1 kWh) and the
    reward at three energies per hour.  An episode cannot pin this function: the env hands it `hour` as a float with quarter
    hours (sustaindc_env.py:679), which the table's integer keys only match on the full hour (KeyError otherwise)."""
    from utils import reward_creator
    hours = np.arange(24)
    energies = np.array([1.0, 331.25, 612.5])
    rew = np.array([[reward_creator.tou_reward({"bat_total_energy_with_battery_KWh": float(e), "energy_usage": float(e),
                                                 "hour": int(h)}) for h in hours] for e in energies])
    assert reward_creator.get_reward_method("tou_reward") is reward_creator.tou_reward
    return {"hours": hours, "energies": energies, "reward": rew, "price": -rew[0]}


def _find_early_seed():
    """Seed for which (day 0, hour < 4) -> cursor < 16; uses python `random` exactly like reset()."""
    for s in range(1000):
        random.seed(s)
        d = random.randint(0, 0)
        h = random.randint(0, 23)
        if d == 0 and h < 2:
            return s
    raise RuntimeError


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--extras", action="store_true", help="with no --only: also regenerate weather_resets / harl_ny_n4")
    args = ap.parse_args()
    _install_shims()
    extra = {"weather_resets": gen_weather_resets, "harl_ny_n4": gen_harl_layer, "rbc_ny_m7": gen_rbc_episode,
             "tou_prices": gen_tou_prices,
             "harl_ny_n2_concat": lambda: gen_harl_layer(n_envs=2, n_steps=120, seed=23, nonoverlapping=False)}
    for name, fn in extra.items():
        if args.only == name or (args.only is None and args.extras):
            out = fn()
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")
    if args.only in extra:
        return
    SPECS["ny_m0_early_cursor"]["seed"] = _find_early_seed()
    for name, spec in SPECS.items():
        if args.only and name != args.only:
            continue
        if "dc_config_file" in spec and not os.path.exists(spec["dc_config_file"]):
            print(f"skip {name}: {spec['dc_config_file']} missing")
            continue
        out = run_fixture(spec)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
