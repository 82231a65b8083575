"""Host-side init-time constants (dc_config.size_datacenter) and trace processing vs the golden vectors,
and the oracle's own sizing routine vs the same vectors.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from dc_rl_amd import dc_config, traces
from oracle import pyoracle as po
from tests.conftest import GOLDEN_DIR, golden_names

LOC_OF = lambda d: str(d["meta_location"])


def _sized(d):
    ci_loc, _ = traces.obtain_paths(LOC_OF(d))
    return dc_config.size_datacenter(str(d["meta_dc_config"]), float(d["meta_capacity_mw"]),
                                     traces.max_ambient_for_sizing(ci_loc))


@pytest.mark.parametrize("name", golden_names())
def test_sizing_matches_reference(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    p = _sized(d)
    for k in ("rack_n", "rack_full", "rack_idle", "rack_supply", "rack_return"):
        np.testing.assert_array_equal(p[k], d["static_" + k])
    for k in ("m_cpu", "c_cpu", "rs_cpu", "m_fan", "c_fan", "rs_fan", "itfan_ref_p", "itfan_ref_v_ratio",
              "it_fan_full_load_v", "c_air", "rho_air", "crac_supply_pu", "min_temp", "max_temp"):
        assert p[k] == float(d["static_" + k]), k
    for k in ("ctafr", "ct_fan_ref_p", "bat_capacity", "power_lb_kW", "power_ub_kW"):
        np.testing.assert_allclose(p[k], float(d["static_" + k]), rtol=1e-13, err_msg=k)
    r = p["ranges"]
    np.testing.assert_allclose(r["Zone Air Temperature(West Zone)"], d["static_range_zone_air"], rtol=1e-13)
    np.testing.assert_allclose(r["Facility Total HVAC Electricity Demand Rate(Whole Building)"], d["static_range_hvac"], rtol=1e-13)
    np.testing.assert_allclose(r["Facility Total Electricity Demand Rate(Whole Building)"], d["static_range_total"], rtol=1e-13)
    np.testing.assert_allclose(r["Facility Total Building Electricity Demand Rate(Whole Building)"], d["static_range_it"], rtol=1e-13)
    assert traces.get_init_day(int(d["meta_month"])) == int(d["static_init_day"])
    # bat dcload bounds: sustaindc_env.py:158-160
    np.testing.assert_allclose(p["power_ub_kW"] / 4, float(d["static_bat_dcload_max"]), rtol=1e-13)
    np.testing.assert_allclose(p["power_lb_kW"] / 4, float(d["static_bat_dcload_min"]), rtol=1e-13)


@pytest.mark.parametrize("name", ["ny_m6_random", "ga_m4_idle_halfmw", "ny_m5_r25", "az_m7_stpt_saw"])
def test_oracle_sizing_matches_reference(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    p = po.params_from_fixture(d)
    p.ctafr = 0.0
    p.ct_fan_ref_p = 0.0
    p.bat_capacity = 0.0
    rg = np.zeros(8)
    ci_loc, _ = traces.obtain_paths(LOC_OF(d))
    po.lib().sdco_size_datacenter(C.byref(p), traces.max_ambient_for_sizing(ci_loc), rg.ctypes.data_as(C.POINTER(C.c_double)))
    np.testing.assert_allclose(p.ctafr, float(d["static_ctafr"]), rtol=1e-13)
    np.testing.assert_allclose(p.ct_fan_ref_p, float(d["static_ct_fan_ref_p"]), rtol=1e-13)
    np.testing.assert_allclose(p.bat_capacity, float(d["static_bat_capacity"]), rtol=1e-13)
    np.testing.assert_allclose(rg[0:2], d["static_range_zone_air"], rtol=1e-13)
    np.testing.assert_allclose(rg[2:4], d["static_range_hvac"], rtol=1e-13)
    np.testing.assert_allclose(rg[4:6], d["static_range_total"], rtol=1e-13)
    np.testing.assert_allclose(rg[6:8], d["static_range_it"], rtol=1e-13)


REF_DATA = "/root/reference/data"


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference data tree not present (GPU box)")
@pytest.mark.parametrize("name", ["ny_m6_random", "ca_m3_defer_drain", "wa_m11_bat_cycle"])
def test_trace_loader_matches_reference_tables(name):
    """On-disk formats -> tables: W, C and the derived NC are bit-identical to what
    the reference's managers held; T / WB are only comparable before noise, which the fixtures do not hold."""
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    tb = traces.load_tables(REF_DATA, LOC_OF(d))
    lo = int(d["ep0_win_lo"])
    n = len(d["ep0_W"])
    np.testing.assert_array_equal(tb["W"][lo:lo + n], d["ep0_W"])
    np.testing.assert_array_equal(tb["C"][lo:lo + n], d["ep0_C"])
    c0 = int(d["ep0_cursor0"])
    cmin, cmax = tb["C"][c0:c0 + 2880].min(), tb["C"][c0:c0 + 2880].max()
    assert cmin == float(d["ep0_ci_min30"]) and cmax == float(d["ep0_ci_max30"])
    np.testing.assert_array_equal((tb["C"][lo:lo + n] - cmin) / (cmax - cmin), d["ep0_NC"])


def test_synthetic_tables_shape_and_ranges():
    tb = traces.synthetic_tables("ny", seed=0)
    for k in ("W", "C", "T", "WB"):
        assert tb[k].shape == (traces.TABLE_LEN,) and np.isfinite(tb[k]).all()
    assert 0.0 <= tb["W"].min() and tb["W"].max() <= 1.0
    assert tb["C"].min() >= 0
    assert (tb["WB"] <= tb["T"] + 1e-6).all()
    tb2 = traces.synthetic_tables("ny", seed=0)
    np.testing.assert_array_equal(tb["W"], tb2["W"])


def test_constant_division_shortcut_is_exact(tmp_path):
    """SDC_DIV_CONST (csrc/sdc_device.hpp): the 3-instruction division by a compile-time constant must give the IEEE
    quotient for every constant the kernels use (3e6 samples each incl. near-all-ones / near-power-of-two significands;
    1e8 each were run once for DESIGN.md)."""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aux", "div_const_check.c")
    exe = str(tmp_path / "div_const_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe, "3000000"], check=True, capture_output=True, text=True).stdout.strip()
    assert out == "0"
