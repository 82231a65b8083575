"""Wet-bulb temperature from dry-bulb / relative humidity / pressure (SI units).

The reference builds its wet-bulb trace with the third-party package
PsychroLib==2.5.0 (`/root/reference/requirements.txt:53`), calling
`psy.GetTWetBulbFromRelHum(t, rh/100, p)` once per hourly EPW row
(`/root/reference/utils/managers.py:530`).  PsychroLib is not vendored in the
reference tree and is not installed in this image, so this module restates its
*published* algorithm (ASHRAE Handbook - Fundamentals 2017, ch. 1: eqns 5, 6,
22, 33, 35 and the bisection / Newton-Raphson solvers PsychroLib documents).

Pinned to PUBLISHED values, not to captured outputs: PsychroLib cannot be run here, so there is no output of the
reference's own call to compare with; the routine is checked against every public known answer instead --
PsychroLib 2.5.0's own SI test values (incl. the sub-zero branch and the 1e-7 humidity-ratio clamp) and the ASHRAE
2017 tables / worked example they quote, within the tolerances stated there
(`tests/test_weather_fixture.py::test_wet_bulb_known_answers`).  It only feeds the `WB` *input table* of the step
(water usage, `/root/reference/envs/datacenter.py:343`), so step parity does not depend on it.
"""
from __future__ import annotations

import math

_ZERO_C_K = 273.15
_TRIPLE_POINT_WATER = 0.01
_MIN_HUM_RATIO = 1e-7
_TOL = 0.001
_MAX_ITER = 100


def sat_vap_pres(t_dry: float) -> float:
    """Saturation vapour pressure (Pa) over ice / liquid water (ASHRAE eqns 5, 6)."""
    t = t_dry + _ZERO_C_K
    if t_dry <= _TRIPLE_POINT_WATER:
        ln_pws = (-5.6745359e3 / t + 6.3925247 - 9.677843e-3 * t + 6.2215701e-7 * t * t
                  + 2.0747825e-9 * t ** 3 - 9.484024e-13 * t ** 4 + 4.1635019 * math.log(t))
    else:
        ln_pws = (-5.8002206e3 / t + 1.3914993 - 4.8640239e-2 * t + 4.1764768e-5 * t * t
                  - 1.4452093e-8 * t ** 3 + 6.5459673 * math.log(t))
    return math.exp(ln_pws)


def _dln_sat_vap_pres(t_dry: float) -> float:
    t = t_dry + _ZERO_C_K
    if t_dry <= _TRIPLE_POINT_WATER:
        return (5.6745359e3 / (t * t) - 9.677843e-3 + 2 * 6.2215701e-7 * t
                + 3 * 2.0747825e-9 * t * t - 4 * 9.484024e-13 * t ** 3 + 4.1635019 / t)
    return (5.8002206e3 / (t * t) - 4.8640239e-2 + 2 * 4.1764768e-5 * t
            - 3 * 1.4452093e-8 * t * t + 6.5459673 / t)


def sat_hum_ratio(t_dry: float, pressure: float) -> float:
    pws = sat_vap_pres(t_dry)
    return max(0.621945 * pws / (pressure - pws), _MIN_HUM_RATIO)


def hum_ratio_from_rel_hum(t_dry: float, rel_hum: float, pressure: float) -> float:
    vap = rel_hum * sat_vap_pres(t_dry)
    return max(0.621945 * vap / (pressure - vap), _MIN_HUM_RATIO)


def t_dew_point_from_vap_pres(t_dry: float, vap_pres: float) -> float:
    """Newton-Raphson on ln(Pws), bounded to the validity range [-100, 200] C."""
    lo, hi = -100.0, 200.0
    t_dew = t_dry
    ln_vp = math.log(vap_pres)
    for _ in range(_MAX_ITER):
        t_iter = t_dew
        ln_vp_iter = math.log(sat_vap_pres(t_iter))
        d_ln_vp = _dln_sat_vap_pres(t_iter)
        t_dew = t_iter - (ln_vp_iter - ln_vp) / d_ln_vp
        t_dew = min(max(t_dew, lo), hi)
        if abs(t_dew - t_iter) <= _TOL:
            break
    return min(t_dew, t_dry)


def hum_ratio_from_t_wet_bulb(t_dry: float, t_wet: float, pressure: float) -> float:
    ws_star = sat_hum_ratio(t_wet, pressure)
    if t_wet >= 0:
        w = ((2501.0 - 2.326 * t_wet) * ws_star - 1.006 * (t_dry - t_wet)) / \
            (2501.0 + 1.86 * t_dry - 4.186 * t_wet)
    else:
        w = ((2830.0 - 0.24 * t_wet) * ws_star - 1.006 * (t_dry - t_wet)) / \
            (2830.0 + 1.86 * t_dry - 2.1 * t_wet)
    return max(w, _MIN_HUM_RATIO)


def t_wet_bulb_from_hum_ratio(t_dry: float, hum_ratio: float, pressure: float) -> float:
    w = max(hum_ratio, _MIN_HUM_RATIO)
    vap = pressure * w / (0.621945 + w)
    t_dew = t_dew_point_from_vap_pres(t_dry, vap)
    sup, inf = t_dry, t_dew
    t_wet = 0.5 * (inf + sup)
    it = 1
    while (sup - inf) > _TOL:
        w_star = hum_ratio_from_t_wet_bulb(t_dry, t_wet, pressure)
        if w_star > w:
            sup = t_wet
        else:
            inf = t_wet
        t_wet = 0.5 * (sup + inf)
        it += 1
        if it >= _MAX_ITER:
            break
    return t_wet


def t_wet_bulb_from_rel_hum(t_dry: float, rel_hum: float, pressure: float) -> float:
    """Counterpart of `psy.GetTWetBulbFromRelHum` (rel_hum in [0, 1], pressure in Pa)."""
    return t_wet_bulb_from_hum_ratio(t_dry, hum_ratio_from_rel_hum(t_dry, rel_hum, pressure), pressure)
