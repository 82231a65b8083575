"""Trace tables for the SustainDC step: on-disk formats -> struct-of-arrays float64 tables of 35 040 samples.

Restates the *construction* part of the reference's managers (`/root/reference/utils/managers.py`):
  * workload  : `Workload_Manager.__init__` :165-191 (hourly CSV column `cpu_load`, first 8760 rows,
                hourly -> 15 min by `np.interp` on `linspace(0, 8760, 35040)`), then the per-reset but
                deterministic `scale_array` :220-244 + 16-tap 'same' moving average :199-208, :268-271
  * carbon    : `CI_Manager.__init__` :342-380 (column `avg_CI`, NaN -> mean, same interpolation), clip >= 0 :417
  * weather   : `Weather_Manager.__init__` :518-562 (EPW: 8 header rows, column 6 dry bulb C, 8 RH %, 9 pressure Pa;
                wet bulb per hourly row, then interpolation)
The per-episode parts (noise, roll, clip, 30-day normalisation) run on the GPU (csrc/sdc_reset.hip).

When the reference's data files are not available (the GPU box has no /root/reference and no network),
`synthetic_tables` produces tables with the value distributions SURVEY.md section 8(d) lists.
"""
from __future__ import annotations

import csv
import os
from typing import Dict, Optional

import numpy as np

from . import psychro

TABLE_LEN = 35040
HOURS = 8760

# location -> (carbon-intensity file stem, EPW file) -- utils/utils_cf.py:11-39
_LOCATIONS = [
    ("az", "AZ", "USA_AZ_Phoenix-Sky.Harbor.epw"),
    ("ca", "CA", "USA_CA_San.Jose-Mineta.epw"),
    ("ga", "GA", "USA_GA_Atlanta-Hartsfield-Jackson.epw"),
    ("il", "IL", "USA_IL_Chicago.OHare.epw"),
    ("ny", "NY", "USA_NY_New.York-LaGuardia.epw"),
    ("tx", "TX", "USA_TX_Dallas-Fort.Worth.epw"),
    ("va", "VA", "USA_VA_Leesburg.Exec.epw"),
    ("wa", "WA", "USA_WA_Seattle-Tacoma.epw"),
]

# first day of each month in a 365-day year (utils/utils_cf.py:56-77 evaluates to this)
MONTH_INIT_DAY = [0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334]


def obtain_paths(location: str):
    """Same substring matching (and order) as utils/utils_cf.py:11-39."""
    loc = location.lower()
    for key, ci, wea in _LOCATIONS:
        if key in loc:
            return [ci, wea]
    raise ValueError(f"Location not found, please define the location {location}")


def get_init_day(start_month: int = 0) -> int:
    assert 0 <= start_month <= 11, "start_month should be between 0 and 11 (inclusive, 0-based, 0=January, 11=December)."
    return MONTH_INIT_DAY[start_month]


def max_ambient_for_sizing(ci_location: str) -> float:
    """utils/make_envs_pyenv.py:149-157 (chiller sizing ambient by location code)."""
    loc = ci_location.lower()
    if "ny" in loc:
        return 30.0
    if "az" in loc:
        return 50.0
    if "wa" in loc:
        return 20.0
    return 50.0


def hourly_to_quarter(x: np.ndarray) -> np.ndarray:
    """np.interp(linspace(0, n, 4n), range(n), x) -- managers.py:183-185."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    return np.interp(np.linspace(0, n, n * 4), np.arange(n), x)


def process_workload(cpu_hourly: np.ndarray, timezone_shift: int = 0) -> np.ndarray:
    """Hourly cpu_load -> the `cpu_smooth` table every reset reproduces (managers.py:183-191, :268-271)."""
    assert len(cpu_hourly) == HOURS, "The number of data points in the workload data is not one year data=24*365=8760."
    w = hourly_to_quarter(cpu_hourly)
    w = np.roll(w, -1 * timezone_shift * 4)
    p5, p95 = np.percentile(w, 5), np.percentile(w, 95)
    w = 0.2 + ((w - p5) * (0.8 - 0.2) / (p95 - p5))
    w = np.clip(w, 0, 1)
    return np.convolve(w, np.ones(16) / 16, mode="same")


def process_carbon(ci_hourly: np.ndarray, timezone_shift: int = 0) -> np.ndarray:
    assert len(ci_hourly) == HOURS, "The number of data points in the carbon intensity data is not one year data=24*365=8760."
    c = np.asarray(ci_hourly, dtype=np.float64)
    if np.isnan(c).any():
        c = np.nan_to_num(c, nan=np.nanmean(c))
    c = hourly_to_quarter(c)
    c = np.roll(c, -1 * timezone_shift * 4)
    return np.clip(c, 0, None)


def process_weather(t_hourly, rh_hourly, p_hourly, timezone_shift: int = 0):
    """-> (dry bulb, wet bulb) 35 040-sample tables BEFORE per-episode noise (managers.py:526-558)."""
    t_hourly = np.asarray(t_hourly, dtype=np.float64)
    wb_hourly = np.array([psychro.t_wet_bulb_from_rel_hum(float(t), float(rh) / 100.0, float(p))
                          for t, rh, p in zip(t_hourly, rh_hourly, p_hourly)])
    t = hourly_to_quarter(t_hourly)
    wb = hourly_to_quarter(wb_hourly)
    return np.roll(t, -1 * timezone_shift * 4), np.roll(wb, -1 * timezone_shift * 4)


def _read_csv_column(path: str, column: str, limit: int) -> np.ndarray:
    """Column of a CSV as float64.  The reference reads with pandas, whose default ("fast") float parser is not
    always correctly rounded; use pandas when it is importable so tables match bit for bit."""
    try:
        import pandas as pd
        return pd.read_csv(path)[column].values[:limit].astype(float)
    except ImportError:  # pragma: no cover - pandas ships in the target image
        out = []
        with open(path, newline="") as f:
            r = csv.reader(f)
            idx = next(r).index(column)
            for row in r:
                if len(out) >= limit:
                    break
                out.append(float(row[idx]) if row[idx] != "" else np.nan)
        return np.array(out, dtype=np.float64)


def _read_epw(path: str):
    """EPW: 8 header rows; column 6 dry bulb (C), 8 relative humidity (%), 9 pressure (Pa) -- managers.py:521-528."""
    try:
        import pandas as pd
        v = pd.read_csv(path, skiprows=8, header=None).values
        return v[:, 6].astype(float), v[:, 8].astype(float), v[:, 9].astype(float)
    except ImportError:  # pragma: no cover
        t, rh, p = [], [], []
        with open(path, newline="") as f:
            r = csv.reader(f)
            for _ in range(8):
                next(r)
            for row in r:
                if row:
                    t.append(float(row[6]))
                    rh.append(float(row[8]))
                    p.append(float(row[9]))
        return np.array(t), np.array(rh), np.array(p)


def load_tables(data_root: str, location: str, workload_file: str = "Alibaba_CPU_Data_Hourly_1.csv",
                timezone_shift: int = 0) -> Dict[str, np.ndarray]:
    """Read the reference's on-disk formats (a `data/` tree laid out like the reference's) into tables."""
    ci_loc, wea_file = obtain_paths(location)
    w = _read_csv_column(os.path.join(data_root, "Workload", workload_file), "cpu_load", HOURS)
    c = _read_csv_column(os.path.join(data_root, "CarbonIntensity", f"{ci_loc}_NG_&_avgCI.csv"), "avg_CI", HOURS)
    t, rh, p = _read_epw(os.path.join(data_root, "Weather", wea_file))
    T, WB = process_weather(t, rh, p, timezone_shift)
    return {"W": process_workload(w, timezone_shift), "C": process_carbon(c, timezone_shift), "T": T, "WB": WB}


# ---------------------------------------------------------------------------------------------- synthetic
_PROFILES = {
    # mean T (C), seasonal amplitude, diurnal amplitude, CI mean, CI amplitude
    "ny": (13.0, 12.0, 4.0, 265.0, 55.0),
    "az": (24.0, 11.0, 6.5, 330.0, 60.0),
    "ca": (16.0, 6.0, 5.0, 230.0, 70.0),
    "wa": (11.0, 7.5, 4.0, 120.0, 45.0),
    "tx": (20.0, 10.0, 5.5, 390.0, 50.0),
    "il": (10.5, 14.0, 4.5, 420.0, 60.0),
    "ga": (17.5, 9.5, 5.0, 400.0, 45.0),
    "va": (13.5, 11.5, 5.0, 350.0, 50.0),
}


def synthetic_tables(location: str = "ny", seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic year of traces with the shapes / ranges SURVEY.md 8(d) lists (W in ~[0.1, 0.9] smooth;
    C ~ 155..380 gCO2/kWh for NY; T ~ -10..36 C raw; WB <= T).  Goes through the same processing functions as
    real data so the tables have the reference's smoothness (hourly data interpolated to 15 minutes)."""
    key = next((k for k in _PROFILES if k in location.lower()), "ny")
    tm, ta, td, cm, ca = _PROFILES[key]
    rng = np.random.default_rng(seed + 7919 * sorted(_PROFILES).index(key))
    h = np.arange(HOURS)
    doy = h / 24.0
    hod = h % 24
    # slow AR(1) weather fronts
    def ar1(sigma, rho):
        e = rng.normal(0, sigma, HOURS)
        x = np.zeros(HOURS)
        for i in range(1, HOURS):
            x[i] = rho * x[i - 1] + e[i]
        return x
    t = tm - ta * np.cos(2 * np.pi * (doy - 20) / 365.0) - td * np.cos(2 * np.pi * (hod - 3) / 24.0) + ar1(0.55, 0.985)
    rh = np.clip(62 - 1.2 * (t - tm) + ar1(1.6, 0.97), 8, 100)
    p = 101325.0 + ar1(18.0, 0.99)
    ci = cm + ca * (0.6 * np.cos(2 * np.pi * (hod - 19) / 24.0) + 0.4 * np.cos(2 * np.pi * (doy - 200) / 365.0)) + ar1(4.0, 0.96)
    ci = np.clip(ci, 20, None)
    wk = ((h // 24) % 7 < 5).astype(np.float64)
    cpu = 0.42 + 0.16 * np.sin(2 * np.pi * (hod - 9) / 24.0) * (0.6 + 0.4 * wk) + 0.05 * wk + ar1(0.012, 0.9)
    cpu = np.clip(cpu, 0.02, 0.98)
    T, WB = process_weather(t, rh, p)
    return {"W": process_workload(cpu), "C": process_carbon(ci), "T": T, "WB": WB}


_cache: Dict[tuple, Dict[str, np.ndarray]] = {}


def get_tables(location: str, workload_file: str = "Alibaba_CPU_Data_Hourly_1.csv", timezone_shift: int = 0,
               data_root: Optional[str] = None, seed: int = 0) -> Dict[str, np.ndarray]:
    """Real tables when a reference-style data tree is available (env SUSTAINDC_DATA or `data_root`), else
    synthetic.  The returned dict carries `source` = 'files' | 'synthetic'."""
    root = data_root or os.environ.get("SUSTAINDC_DATA")
    key = (location.lower(), workload_file, timezone_shift, root, seed)
    if key in _cache:
        return _cache[key]
    if root and os.path.isdir(root):
        tb = load_tables(root, location, workload_file, timezone_shift)
        tb["source"] = "files"
    else:
        tb = synthetic_tables(location, seed)
        tb["source"] = "synthetic"
    _cache[key] = tb
    return tb
