"""Data-centre parameters: JSON -> rack table -> sized HVAC / battery constants (host side, once).

Restates the init-time path of the reference:
  * `DC_Config._setup_config`      utils/dc_config_reader.py:39-145  (JSON keys, per-rack CPU lists)
  * `Rack.__init__` power cap       envs/datacenter.py:54-76
  * `CPU.cpu_curve1/itfan_curve2`   envs/datacenter.py:31-49
  * `chiller_sizing`                envs/datacenter.py:476-529
  * the 8 x 11 sizing sweep, ranges and battery sizing   utils/make_envs_pyenv.py:139-218

Rack order is the JSON order.  (The reference builds the per-rack lists in a ThreadPool and collects them
with `as_completed`, so its order is non-deterministic -- utils/dc_config_reader.py:100-105.)
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict

import numpy as np

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

MIN_TEMP = 15.0   # utils/make_envs_pyenv.py:125
MAX_TEMP = 21.6   # utils/make_envs_pyenv.py:126
INIT_SETPOINT = 18  # utils/make_envs_pyenv.py:124


def resolve_config_path(dc_config_file: str) -> str:
    """Absolute paths are used as-is; bare names resolve against this package's configs/ (the reference
    resolves against its utils/ directory, utils/dc_config_reader.py:19)."""
    if os.path.isabs(dc_config_file):
        return dc_config_file
    return os.path.join(CONFIG_DIR, dc_config_file)


class DCConfig:
    """Attribute bag with the reference's DC_Config names (NUM_RACKS, RACK_SUPPLY_APPROACH_TEMP_LIST, C_AIR, ...)."""

    def __init__(self, dc_config_file: str = "dc_config.json", datacenter_capacity_mw: float = 1):
        self.config_path = resolve_config_path(dc_config_file)
        self.datacenter_capacity_mw = datacenter_capacity_mw
        with open(self.config_path, "r") as f:
            j = json.load(f)
        g, hv, sv = j["data_center_configuration"], j["hvac_configuration"], j["server_characteristics"]
        self.NUM_ROWS = g["NUM_ROWS"]
        self.NUM_RACKS_PER_ROW = g["NUM_RACKS_PER_ROW"]
        self.NUM_RACKS = self.NUM_ROWS * self.NUM_RACKS_PER_ROW
        self.TOTAM_MAX_PWR = self.datacenter_capacity_mw * 1e6
        self.MAX_W_PER_RACK = int(self.TOTAM_MAX_PWR / self.NUM_RACKS)
        self.RACK_SUPPLY_APPROACH_TEMP_LIST = list(g["RACK_SUPPLY_APPROACH_TEMP_LIST"])
        self.RACK_RETURN_APPROACH_TEMP_LIST = list(g["RACK_RETURN_APPROACH_TEMP_LIST"])
        self.CPUS_PER_RACK = g["CPUS_PER_RACK"]
        self.DEFAULT_SERVER_POWER_CHARACTERISTICS = sv["DEFAULT_SERVER_POWER_CHARACTERISTICS"]
        assert len(self.DEFAULT_SERVER_POWER_CHARACTERISTICS) == self.NUM_RACKS, \
            "DEFAULT_SERVER_POWER_CHARACTERISTICS should be of length as NUM_RACKS"
        if not (len(self.RACK_SUPPLY_APPROACH_TEMP_LIST) >= self.NUM_RACKS
                and len(self.RACK_RETURN_APPROACH_TEMP_LIST) >= self.NUM_RACKS):
            raise ValueError("approach-temperature lists are shorter than NUM_RACKS")
        self.HP_PROLIANT = sv["HP_PROLIANT"]
        self.CPU_POWER_RATIO_LB = sv["CPU_POWER_RATIO_LB"]
        self.CPU_POWER_RATIO_UB = sv["CPU_POWER_RATIO_UB"]
        self.IT_FAN_AIRFLOW_RATIO_LB = sv["IT_FAN_AIRFLOW_RATIO_LB"]
        self.IT_FAN_AIRFLOW_RATIO_UB = sv["IT_FAN_AIRFLOW_RATIO_UB"]
        self.IT_FAN_FULL_LOAD_V = sv["IT_FAN_FULL_LOAD_V"]
        self.ITFAN_REF_V_RATIO, self.ITFAN_REF_P = sv["ITFAN_REF_V_RATIO"], sv["ITFAN_REF_P"]
        self.INLET_TEMP_RANGE = sv["INLET_TEMP_RANGE"]
        self.C_AIR = hv["C_AIR"]
        self.RHO_AIR = hv["RHO_AIR"]
        self.CRAC_SUPPLY_AIR_FLOW_RATE_pu = hv["CRAC_SUPPLY_AIR_FLOW_RATE_pu"]
        self.CRAC_REFRENCE_AIR_FLOW_RATE_pu = hv["CRAC_REFRENCE_AIR_FLOW_RATE_pu"]
        self.CRAC_FAN_REF_P = hv["CRAC_FAN_REF_P"]
        self.CHILLER_COP = hv["CHILLER_COP_BASE"]
        self.CW_PRESSURE_DROP = hv["CW_PRESSURE_DROP"]
        self.CW_WATER_FLOW_RATE = hv["CW_WATER_FLOW_RATE"]
        self.CW_PUMP_EFFICIENCY = hv["CW_PUMP_EFFICIENCY"]
        self.CHILLER_COP_K = hv["CHILLER_COP_K"]
        self.CHILLER_COP_T_NOMINAL = hv["CHILLER_COP_T_NOMINAL"]
        self.CT_FAN_REF_P = hv["CT_FAN_REF_P"]
        self.CT_REFRENCE_AIR_FLOW_RATE = hv["CT_REFRENCE_AIR_FLOW_RATE"]
        self.CT_PRESSURE_DROP = hv["CT_PRESSURE_DROP"]
        self.CT_WATER_FLOW_RATE = hv["CT_WATER_FLOW_RATE"]
        self.CT_PUMP_EFFICIENCY = hv["CT_PUMP_EFFICIENCY"]

    def rack_table(self) -> Dict[str, np.ndarray]:
        """Per-rack {n, full, idle}: every CPU of a rack is identical (dc_config_reader.py:97); CPUs are added
        until the running full-load sum reaches MAX_W_PER_RACK, the one that reaches it is dropped
        (envs/datacenter.py:67-74)."""
        n, full, idle = [], [], []
        for fl, il in ((c[0], c[-1]) for c in self.DEFAULT_SERVER_POWER_CHARACTERISTICS):
            load, cnt = 0, 0
            for _ in range(int(self.CPUS_PER_RACK)):
                load += fl
                cnt += 1
                if load >= self.MAX_W_PER_RACK:
                    cnt -= 1
                    break
            n.append(cnt)
            full.append(fl)
            idle.append(il)
        R = self.NUM_RACKS
        return {"rack_n": np.array(n, dtype=np.float64), "rack_full": np.array(full, dtype=np.float64),
                "rack_idle": np.array(idle, dtype=np.float64),
                "rack_supply": np.array(self.RACK_SUPPLY_APPROACH_TEMP_LIST[:R], dtype=np.float64),
                "rack_return": np.array(self.RACK_RETURN_APPROACH_TEMP_LIST[:R], dtype=np.float64)}


def chiller_power(max_cooling_cap: float, load: float, ambient_temp: float) -> float:
    """envs/datacenter.py:356-429 (EnergyPlus electric-chiller curves)."""
    min_plr, max_plr = 0.05, 1.0
    delta_temp = (ambient_temp - 35.0) / 2.778 - (6.67 - 35.0)
    cap_rat = 0.94483600 + -0.05700880 * delta_temp + 0.00185486 * delta_temp ** 2
    avail = max_cooling_cap * cap_rat if cap_rat != 0 else 0
    fpr = 2.333 + -1.975 * cap_rat + 0.6121 * cap_rat ** 2
    plr = max(min_plr, min(load / avail, max_plr)) if avail > 0 else 0
    fflp = 0.03303 + 0.6852 * plr + 0.2818 * plr ** 2
    if avail > 0:
        oper = load / avail if load / avail < min_plr else plr
    else:
        oper = 0.0
    frac = min(1.0, oper / min_plr) if oper < min_plr else 1.0
    power = fflp * fpr * avail / 3.0 * frac
    return power if oper > 0 else 0


def _it_model(p: dict, stpt: float, load_pct: float):
    """Rack model at one (set-point, load): returns (P_it, avg CRAC return temp, mean outlet).
    envs/datacenter.py:250-317 with :157-181 collapsed to n x per-CPU (all CPUs of a rack identical)."""
    sa = np.clip(p["rack_supply"], 3.8, 5.3)
    inlet = sa + stpt
    ratio = ((p["m_cpu"] + 0.05) * inlet + p["c_cpu"]) + p["rs_cpu"] * (load_pct / 100)
    cpu1 = np.maximum(p["rack_idle"], p["rack_full"] * ratio)
    v = (p["m_fan"] * 10 * inlet + p["c_fan"] * 5) + p["rs_fan"] * (load_pct / 20)
    fan1 = p["itfan_ref_p"] * (v / p["itfan_ref_v_ratio"])
    vtot = p["rack_n"] * (p["it_fan_full_load_v"] * v)
    pcpu, pfan = p["rack_n"] * cpu1, p["rack_n"] * fan1
    outlet = inlet + 1.918 * (pcpu + pfan) ** 1.096 / (p["c_air"] * p["rho_air"] * vtot ** 0.824 * 0.526) + -14.01
    if np.any(outlet - inlet < 2):
        raise ValueError("rack outlet-inlet temperature delta < 2 C for this data-centre configuration "
                         "(the reference raises here too: envs/datacenter.py:295-300)")
    p_it = float(sum(pcpu.tolist()) + sum(pfan.tolist()))
    avg_ret = float(sum((p["rack_return"] + outlet).tolist()) / len(outlet))
    return p_it, avg_ret, float(sum(outlet.tolist()) / len(outlet))


def size_datacenter(dc_config_file: str = "dc_config.json", datacenter_capacity_mw: float = 1,
                    max_ambient_temp: float = 30.0) -> dict:
    """-> parameter dict for `SdcEngine.set_dc_params` + the reference's `ranges` / power bounds."""
    cfg = DCConfig(dc_config_file, datacenter_capacity_mw)
    p = dict(cfg.rack_table())
    lo_t, hi_t = cfg.INLET_TEMP_RANGE
    p["m_cpu"] = (cfg.CPU_POWER_RATIO_UB[0] - cfg.CPU_POWER_RATIO_LB[0]) / (hi_t - lo_t)
    p["c_cpu"] = cfg.CPU_POWER_RATIO_UB[0] - p["m_cpu"] * hi_t
    p["rs_cpu"] = cfg.CPU_POWER_RATIO_LB[1] - cfg.CPU_POWER_RATIO_LB[0]
    p["m_fan"] = (cfg.IT_FAN_AIRFLOW_RATIO_UB[0] - cfg.IT_FAN_AIRFLOW_RATIO_LB[0]) / (hi_t - lo_t)
    p["c_fan"] = cfg.IT_FAN_AIRFLOW_RATIO_UB[0] - p["m_fan"] * hi_t
    p["rs_fan"] = cfg.IT_FAN_AIRFLOW_RATIO_LB[1] - cfg.IT_FAN_AIRFLOW_RATIO_LB[0]
    p["itfan_ref_p"] = cfg.ITFAN_REF_P
    p["itfan_ref_v_ratio"] = cfg.ITFAN_REF_V_RATIO
    p["it_fan_full_load_v"] = cfg.IT_FAN_FULL_LOAD_V
    p["c_air"], p["rho_air"] = cfg.C_AIR, cfg.RHO_AIR
    p["crac_supply_pu"] = cfg.CRAC_SUPPLY_AIR_FLOW_RATE_pu
    p["min_temp"], p["max_temp"], p["init_setpoint"] = MIN_TEMP, MAX_TEMP, INIT_SETPOINT
    # chiller_sizing(min_CRAC=15, max_CRAC=21.6, max_ambient): envs/datacenter.py:476-529
    p_it, avg_ret, _ = _it_model(p, MAX_TEMP, 100.0)
    m_sys = cfg.RHO_AIR * cfg.CRAC_SUPPLY_AIR_FLOW_RATE_pu * p_it
    q = m_sys * cfg.C_AIR * max(0.0, avg_ret - MIN_TEMP)
    delta = max(50 - (max_ambient_temp - MIN_TEMP), 1)
    m_air = q / (cfg.C_AIR * delta)
    p["ctafr"] = m_air / cfg.RHO_AIR
    p["ct_fan_ref_p"] = q
    # 8 set-points x 11 loads sweep: utils/make_envs_pyenv.py:166-178
    it, zone = [], []
    for s in range(15, 23):
        for l in range(0, 110, 10):
            a, _, c = _it_model(p, s, l)
            it.append(a)
            zone.append(c)
    chiller_max = chiller_power(q, max(it), max_ambient_temp)
    max_dc_power_w = 1.1 * max(it) + 1.1 * q + 1.1 * chiller_max
    max_dc_energy = (max_dc_power_w / 4) * (4 * 1) / 1e6
    p["bat_capacity"] = max_dc_energy
    ranges = {
        "sinhour": [-1.0, 1.0], "coshour": [-1.0, 1.0], "sindayOTY": [-1.0, 1.0], "cosdayOTY": [-1.0, 1.0],
        "hour": [0.0, 23.0], "dayOTY": [1.0, 366.0],
        "Site Outdoor Air Drybulb Temperature(Environment)": [-10.0, 40.0],
        "Zone Thermostat Cooling Setpoint Temperature(West Zone)": [15.0, 30.0],
        "Zone Air Temperature(West Zone)": [0.9 * min(zone), 1.1 * max(zone)],
        "Facility Total HVAC Electricity Demand Rate(Whole Building)": [0.0, 1.1 * q + 1.1 * chiller_max],
        "Facility Total Electricity Demand Rate(Whole Building)": [0.9 * min(it), 1.1 * max(it) + 1.1 * q + 1.1 * chiller_max],
        "Facility Total Building Electricity Demand Rate(Whole Building)": [0.9 * min(it), 1.1 * max(it)],
        "cpuUsage": [0.0, 1.0], "carbonIntensity": [0.0, 1000.0],
        "max_battery_energy_Mwh": max_dc_energy,
    }
    p["ranges"] = ranges
    hv = ranges["Facility Total HVAC Electricity Demand Rate(Whole Building)"]
    itr = ranges["Facility Total Building Electricity Demand Rate(Whole Building)"]
    p["power_lb_kW"] = (itr[0] + hv[0]) / 1e3   # envs/dc_gym.py:86-87
    p["power_ub_kW"] = (itr[1] + hv[1]) / 1e3
    p["max_dc_pw"] = hv[1] + itr[1]
    p["dc_config"] = cfg
    return p
