"""Factory surface of the reference's `utils/make_envs_pyenv.py`, same names and arguments.

In the reference these build three scalar Gymnasium sub-environments whose `step()` is Python.  Here the
dynamics of all three run inside the HIP kernels, so the factories return light *parameter objects* that carry
exactly the attributes `SustainDC.__init__` reads from them (sustaindc_env.py:148-191): spaces, `ranges`,
`power_lb_kW / power_ub_kW`, battery bounds -- plus the sized parameter dict the engine consumes.
"""
from __future__ import annotations

import numpy as np

from . import dc_config, traces
from .spaces import Box, Discrete


class LoadShiftingParams:
    """Counterpart of `CarbonLoadEnv` (envs/carbon_ls.py:8-60): spaces and constants only."""

    def __init__(self, n_vars_ci=4, flexible_workload_ratio=0.2, n_vars_energy=0, n_vars_battery=1, test_mode=False,
                 queue_max_len=500):
        assert flexible_workload_ratio < 0.9, "flexible_workload_ratio should be lower than 0.9"
        self.flexible_workload_ratio = flexible_workload_ratio
        self.shiftable_tasks_percentage = flexible_workload_ratio
        self.non_shiftable_tasks_percentage = 1 - flexible_workload_ratio
        self.action_space = Discrete(3)  # 0 defer, 1 do nothing, 2 process the queue (carbon_ls.py:38-40)
        self.observation_space = Box(low=-2.0, high=2.0, shape=(26,), dtype=np.float32)
        self.test_mode = test_mode
        self.time_steps_day = 96
        self.queue_max_len = queue_max_len


class DataCenterParams:
    """Counterpart of `dc_gymenv` (envs/dc_gym.py:11-89)."""

    def __init__(self, sized: dict, observation_space, action_space, action_mapping, min_temp, max_temp):
        self.sized = sized                      # parameter dict for SdcEngine.set_dc_params
        self.DC_Config = sized["dc_config"]
        self.ranges = sized["ranges"]
        self.observation_space = observation_space
        self.action_space = action_space
        self.action_mapping = action_mapping
        self.min_temp = min_temp
        self.max_temp = max_temp
        self.raw_curr_stpt = sized["init_setpoint"]
        self.power_lb_kW = sized["power_lb_kW"]   # dc_gym.py:86-87
        self.power_ub_kW = sized["power_ub_kW"]


class BatteryParams:
    """Counterpart of `BatteryEnvFwd` (envs/bat_env_fwd_view.py:8-52)."""

    def __init__(self, env_config):
        self.observation_space = Box(low=np.float32(-1.0 * np.ones(13)), high=np.float32(1.0 * np.ones(13)))
        self.action_space = Discrete(3)           # 0 charge, 1 discharge, 2 idle (bat_env_fwd_view.py:28)
        self._action_to_direction = {0: "charge", 1: "discharge", 2: "idle"}
        self.max_dc_pw_MW = env_config["max_dc_pw_MW"]
        self.max_bat_cap = env_config["max_bat_cap"]
        self.charging_rate = env_config["charging_rate"]
        self.n_fwd_steps = env_config["n_fwd_steps"]
        self.dcload_max = env_config["dcload_max"]
        self.dcload_min = env_config["dcload_min"]


def make_ls_env(month, n_vars_ci: int = 4, n_vars_energy: int = 4, n_vars_battery: int = 1, queue_max_len: int = 500,
                test_mode=False):
    """utils/make_envs_pyenv.py:19-41.  (`flexible_load` is not forwarded by the reference either, so the
    flexible ratio is the class default 0.2.)"""
    return LoadShiftingParams(n_vars_ci=n_vars_ci, n_vars_energy=n_vars_energy, n_vars_battery=n_vars_battery,
                              queue_max_len=queue_max_len, test_mode=test_mode)


def make_bat_fwd_env(month, max_bat_cap_Mwh: float = 2.0, charging_rate: float = 0.5, max_dc_pw_MW: float = 7.23,
                     dcload_max: float = 2.5, dcload_min: float = 0.1, n_fwd_steps: int = 4):
    """utils/make_envs_pyenv.py:45-73."""
    init_day = traces.get_init_day(month)
    return BatteryParams({"n_fwd_steps": n_fwd_steps, "max_dc_pw_MW": max_dc_pw_MW, "max_bat_cap": max_bat_cap_Mwh,
                          "charging_rate": charging_rate, "start_point": init_day, "dcload_max": dcload_max,
                          "dcload_min": dcload_min})


def make_dc_pyeplus_env(month: int = 1, location: str = "NYIS", dc_config_file: str = "dc_config_file.json",
                        datacenter_capacity_mw: int = 1, max_bat_cap_Mw: float = 2.0, add_cpu_usage: bool = True,
                        add_CI: bool = True, episode_length_in_time=None, use_ls_cpu_load: bool = False,
                        num_sin_cos_vars: int = 4):
    """utils/make_envs_pyenv.py:75-242: chiller / cooling-tower sizing for the location, the 88-point sweep for the
    observation ranges, battery sizing.  Returns (dc_env, max_dc_pw) like the reference."""
    observation_space = Box(low=np.float32(-1.0 * np.ones(14)), high=np.float32(1.0 * np.ones(14)))
    action_mapping = {0: -1, 1: 0, 2: 1}
    action_space = Discrete(len(action_mapping))
    sized = dc_config.size_datacenter(dc_config_file, datacenter_capacity_mw, traces.max_ambient_for_sizing(location))
    sized["ranges"]["batterySoC"] = [0.0, max_bat_cap_Mw * 1e6]
    dc_env = DataCenterParams(sized, observation_space, action_space, action_mapping, dc_config.MIN_TEMP,
                              dc_config.MAX_TEMP)
    return dc_env, sized["max_dc_pw"]
