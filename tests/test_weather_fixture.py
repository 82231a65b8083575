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
    """PsychroLib 2.5.0's own SI test suite pins GetTWetBulbFromRelHum(7 C, 0.61, 100 kPa) = 3.92667 C (abs 1e-3)."""
    assert psychro.t_wet_bulb_from_rel_hum(7.0, 0.61, 100000.0) == pytest.approx(3.92667, abs=1e-3)
    # saturation: wet bulb == dry bulb; dry air: far below
    for t in (-5.0, 10.0, 30.0):
        assert psychro.t_wet_bulb_from_rel_hum(t, 1.0, 101325.0) == pytest.approx(t, abs=2e-3)
        assert psychro.t_wet_bulb_from_rel_hum(t, 0.2, 101325.0) < t - 2.0
