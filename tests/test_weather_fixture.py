"""CPU pins of the weather path against tests/golden/weather_resets.npz (captured from the reference's Weather_Manager,
utils/managers.py:488-628, by tests/golden/gen_golden.py --only weather_resets):
  * the test-side restatement of the reset arithmetic (tests/reset_ref.py: add noise, roll, clip, 30-day min / max) that
    every injected-reset parity test relies on -- roll DIRECTION and the normalisation WINDOW included;
  * the on-disk EPW -> 15-minute dry-bulb table of dc_rl_amd.traces (bit-identical; needs the reference's data tree);
  * the wet-bulb routine against PsychroLib's published known-answer value (PsychroLib itself is not installed: the
    wet-bulb TABLE stays parity-unpinned)."""
import os

import numpy as np
import pytest

from dc_rl_amd import psychro, traces
from tests import reset_ref as RR
from tests.conftest import GOLDEN_DIR

REF_DATA = "/root/reference/data"


def _fx():
    return np.load(os.path.join(GOLDEN_DIR, "weather_resets.npz"))


def test_reset_arithmetic_restatement_matches_reference_resets():
    d = _fx()
    W = int(d["meta_window"])
    for k in range(int(d["meta_cases"])):
        pre = f"case{k}_"
        if int(d[pre + "tz"]) != 0:
            continue       # (needs the shifted table: covered by the loader test below)
        loc = str(d[pre + "loc"])
        noise, roll = RR.coherent_noise_legacy(int(d[pre + "seed"]))
        # the regenerated noise IS the array the reference drew (every 32nd sample kept in the fixture)
        np.testing.assert_allclose(noise[::32], d[pre + "noise_sub32"], rtol=0, atol=1e-12)
        assert roll == int(d[pre + "roll_days"])
        c0 = int(d[pre + "cursor0"])
        assert c0 == int(d[pre + "day"]) * 96 + int(d[pre + "hour"]) * 4
        r = RR.reference_weather_reset(d[loc + "_T"], d[loc + "_WB"], noise, roll, c0, W)
        np.testing.assert_allclose(r["T"], d[pre + "T_win"], rtol=0, atol=1e-11)
        np.testing.assert_allclose(r["WB"], d[pre + "WB_win"], rtol=0, atol=1e-11)
        assert abs(r["t_min"] - float(d[pre + "t_min30"])) <= 1e-11 and abs(r["t_max"] - float(d[pre + "t_max30"])) <= 1e-11
        nt = (r["T"] - r["t_min"]) / (r["t_max"] - r["t_min"])          # managers.py:608
        np.testing.assert_allclose(nt, d[pre + "NT_win"], rtol=0, atol=1e-11)
        # a roll in the wrong direction, or min / max over the wrong span, must be visible at this tolerance
        wrong = np.clip(np.roll(d[loc + "_T"] + noise, -roll * 96), 0, 45)[c0:c0 + W]
        if roll:
            assert np.abs(wrong - d[pre + "T_win"]).max() > 1e-3
        assert float(d[pre + "t_max30"]) >= d[pre + "T_win"][:2880].max() - 1e-12
        assert float(d[pre + "t_max30"]) == pytest.approx(d[pre + "T_win"][:2880].max(), abs=1e-12)


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference data tree not present (GPU box)")
def test_epw_loader_matches_reference_tables():
    d = _fx()
    for loc in (str(x) for x in d["meta_locations"]):
        tb = traces.load_tables(REF_DATA, loc)
        np.testing.assert_array_equal(tb["T"], d[loc + "_T"])         # EPW -> hourly -> 15 min: bit-identical
        # the fixture's wet bulb came through the generator's psychrolib shim, i.e. through dc_rl_amd.psychro itself:
        # this only checks the interpolation / roll plumbing, NOT the wet-bulb routine (parity-unpinned)
        np.testing.assert_array_equal(tb["WB"], d[loc + "_WB"])
        assert (tb["WB"] <= tb["T"] + 1e-9).all()
    for k in range(int(d["meta_cases"])):
        pre = f"case{k}_"
        tz = int(d[pre + "tz"])
        if tz == 0:
            continue
        tb = traces.load_tables(REF_DATA, str(d[pre + "loc"]), timezone_shift=tz)   # np.roll(-tz * 4), managers.py:556
        np.testing.assert_array_equal(tb["T"][::16], d[pre + "T_orig_sub16"])


def test_wet_bulb_known_answers():
    """The wet-bulb routine (dc_rl_amd/psychro.py, the restatement of PsychroLib==2.5.0's published algorithm;
    reference call site utils/managers.py:530) against every public known answer there is for it: the values PsychroLib's
    own SI test suite asserts (tests/test_psychrolib_si.py of the 2.5.0 release, tolerances as stated THERE), and the
    ASHRAE Handbook - Fundamentals 2017 ch. 1 tables and worked example those tests quote.  PsychroLib itself cannot be
    run here (not installed, no network), so these are its published numbers, not outputs captured in this container --
    and the fixtures' own wet bulb came through psychro.py (weather_resets.npz `meta_wb_source`)."""
    P = psychro
    # -- PsychroLib: GetTWetBulbFromRelHum (the call the reference makes), rel 1e-3
    assert P.t_wet_bulb_from_rel_hum(7.0, 0.61, 100000.0) == pytest.approx(3.92667433781955, rel=1e-3)
    # -- PsychroLib: humidity ratio <-> wet bulb, above freezing (rel 3e-4, then abs 1e-3 on the way back) ...
    w = P.hum_ratio_from_t_wet_bulb(30.0, 25.0, 95461.0)
    assert w == pytest.approx(0.0192281274241096, rel=3e-4)
    assert P.t_wet_bulb_from_hum_ratio(30.0, w, 95461.0) == pytest.approx(25.0, abs=1e-3)
    # ... and BELOW freezing (the ice branch of the psychrometer equation and of the saturation pressure)
    w = P.hum_ratio_from_t_wet_bulb(-1.0, -5.0, 95461.0)
    assert w == pytest.approx(0.00120399819933844, rel=3e-4)
    assert P.t_wet_bulb_from_hum_ratio(-1.0, w, 95461.0) == pytest.approx(-5.0, abs=1e-3)
    # humidity ratios below 1e-7 are clamped there
    assert P.t_wet_bulb_from_hum_ratio(-5.0, 1e-9, 95461.0) == P.t_wet_bulb_from_hum_ratio(-5.0, 1e-7, 95461.0)
    # -- ASHRAE Table 3 (saturation vapour pressure, Pa), rel 3e-4 (abs 0.01 Pa at -60 C), both branches
    for t, pws in ((-20, 103.24), (-5, 401.74), (5, 872.6), (25, 3169.7), (50, 12351.3), (100, 101418.0), (150, 476101.4)):
        assert P.sat_vap_pres(t) == pytest.approx(pws, rel=3e-4), t
    assert P.sat_vap_pres(-60) == pytest.approx(1.08, abs=0.01)
    # -- ASHRAE Table 2 (saturation humidity ratio at 101 325 Pa): the table includes the enhancement factor the
    #    ideal-gas relation leaves out -- "agreement is not terrific, up to 2 %" in PsychroLib's own words
    for t, ws in ((-50, 0.0000243), (-20, 0.0006373), (-5, 0.0024863), (5, 0.005425), (25, 0.020173), (50, 0.087516),
                  (85, 0.838105)):
        assert P.sat_hum_ratio(t, 101325.0) == pytest.approx(ws, rel=0.02), t
    # -- dew point <-> vapour pressure round trips (PsychroLib: abs 1e-3), ice and liquid
    assert P.t_dew_point_from_vap_pres(15.0, P.sat_vap_pres(-20.0)) == pytest.approx(-20.0, abs=1e-3)
    assert P.t_dew_point_from_vap_pres(60.0, P.sat_vap_pres(50.0)) == pytest.approx(50.0, abs=1e-3)
    # -- ASHRAE worked example 1 (40 C dry bulb, 20 C wet bulb, sea level): W = 0.0065, dew point 7 C, RH 14 %; and
    #    back from the relative humidity to the wet bulb (abs 0.1, PsychroLib's tolerance)
    w = P.hum_ratio_from_t_wet_bulb(40.0, 20.0, 101325.0)
    assert w == pytest.approx(0.0065, abs=1e-4)
    vap = 101325.0 * w / (0.621945 + w)
    assert P.t_dew_point_from_vap_pres(40.0, vap) == pytest.approx(7.0, abs=0.5)
    rh = vap / P.sat_vap_pres(40.0)
    assert rh == pytest.approx(0.14, abs=0.01)
    assert P.t_wet_bulb_from_rel_hum(40.0, rh, 101325.0) == pytest.approx(20.0, abs=0.1)
    # saturation: wet bulb == dry bulb; dry air: far below
    for t in (-5.0, 10.0, 30.0):
        assert P.t_wet_bulb_from_rel_hum(t, 1.0, 101325.0) == pytest.approx(t, abs=2e-3)
        assert P.t_wet_bulb_from_rel_hum(t, 0.2, 101325.0) < t - 2.0
    assert "psychro.py" in str(_fx()["meta_wb_source"])
