"""Pin the CPU oracle (oracle/sdc_oracle.c) to the golden vectors captured from the Python reference.

Bars: obs identical after the float32 cast up to 1 float32 ulp on
the derived features; rewards / energies within 1e-12 relative (1e-12 absolute where |ref| < 1)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests.conftest import GOLDEN_DIR, golden_names

FAST = [n for n in golden_names() if n not in ("ny_m6_multi16", "ca_m6_30day")]
SLOW = [n for n in golden_names() if n in ("ny_m6_multi16", "ca_m6_30day")]


def _run(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    p = po.params_from_fixture(d)
    env = po.OracleEnv(p)
    env.e.stpt = float(d["init_stpt"])
    keys = [str(k) for k in d["meta_info_keys"]]
    steps = int(d["meta_steps"])
    worst = dict(obs=0.0, rew=0.0, info=0.0)
    for ep in range(int(d["meta_episodes"])):
        pre = f"ep{ep}_"
        lo = int(d[pre + "win_lo"])
        obs0 = env.begin(d[pre + "W"], d[pre + "C"], d[pre + "NC"], d[pre + "T"], d[pre + "WB"], d[pre + "NT"], lo,
                         int(d[pre + "init_day"]), int(d[pre + "init_hour"]), steps)
        assert env.e.cursor == int(d[pre + "cursor0"])
        assert env.e.hist_len == int(d[pre + "hist_len0"])
        assert env.e.stpt == float(d[pre + "stpt0"])
        _cmp_obs(obs0, d[pre + "reset_obs"], worst, f"{name} ep{ep} reset")
        acts, gobs, grew, gdone, ginfo, ghist = (d[pre + k] for k in ("actions", "obs", "rew", "done", "info", "age_hist"))
        for t in range(steps):
            obs, rew, done, info = env.step(acts[t])
            tag = f"{name} ep{ep} t{t}"
            _cmp_obs(obs, gobs[t], worst, tag)
            assert done == int(gdone[t]), tag
            err = np.abs(rew - grew[t]) / np.maximum(1.0, np.abs(grew[t]))
            worst["rew"] = max(worst["rew"], float(err.max()))
            assert err.max() <= 1e-11, (tag, rew, grew[t])
            for j, k in enumerate(keys):
                ref = ginfo[t, j]
                got = info[po.INFO_IDX[k]]
                e = abs(got - ref) / max(1.0, abs(ref))
                worst["info"] = max(worst["info"], e)
                assert e <= 1e-12, (tag, k, got, ref)
            np.testing.assert_allclose(info[po.INFO_IDX["ls_hist0"]:po.INFO_IDX["ls_hist0"] + 5], ghist[t], rtol=0, atol=1e-15)
            assert info[po.INFO_IDX["fault"]] == 0
    return worst


def _cmp_obs(got, ref, worst, tag):
    # float32 values: identical, or within 1 ulp on features that go through summation-order-dependent code,
    # or tiny absolute differences around 0 (polyfit returns ~1e-17 where the closed form returns 0)
    diff = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    tol = np.maximum(np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64) * 1.0, 1e-12)
    bad = diff > tol
    worst["obs"] = max(worst["obs"], float(diff.max()))
    assert not bad.any(), (tag, np.nonzero(bad)[0], got[bad], ref[bad])


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference_episode(name):
    w = _run(name)
    print(name, w)


@pytest.mark.slow
@pytest.mark.parametrize("name", SLOW)
def test_oracle_matches_reference_long(name):
    w = _run(name)
    print(name, w)


def test_oracle_tou_reward_uses_the_references_price_table():
    """utils/reward_creator.py:154-198 tou_reward: the oracle's case 3 against the 24 prices captured from the reference's
    function itself (tests/golden/tou_prices.npz, gen_golden.py gen_tou_prices).  On full hours the oracle's reward IS the
    reference's function of (energy, hour); between them the reference raises KeyError and the oracle keeps the hour's price."""
    tp = np.load(os.path.join(GOLDEN_DIR, "tou_prices.npz"))
    price = tp["price"]
    np.testing.assert_array_equal(tp["reward"], -tp["energies"][:, None] * price[None, :])
    d = np.load(os.path.join(GOLDEN_DIR, "ny_m6_random.npz"))
    scal = {k[len("static_"):]: d[k] for k in d.files if k.startswith("static_") and d[k].ndim == 0}
    scal["reward_method"] = (3, 3, 3)
    p = po.make_params(d["static_rack_n"], d["static_rack_full"], d["static_rack_idle"], d["static_rack_supply"],
                       d["static_rack_return"], scal)
    env = po.OracleEnv(p)
    env.e.stpt = float(d["init_stpt"])
    env.begin(d["ep0_W"], d["ep0_C"], d["ep0_NC"], d["ep0_T"], d["ep0_WB"], d["ep0_NT"], int(d["ep0_win_lo"]),
              int(d["ep0_init_day"]), int(d["ep0_init_hour"]), int(d["meta_steps"]))
    seen = set()
    for t in range(200):
        _, rew, _, info = env.step(d["ep0_actions"][t])
        hour = info[po.INFO_IDX["hour"]]
        e = info[po.INFO_IDX["bat_total_energy_with_battery_KWh"]]
        h = int(hour) % 24
        seen.add(h)
        np.testing.assert_array_equal(rew, np.full(3, -1.0 * e * price[h]))
        assert env.e.hist_len == 0          # only default_ls_reward appends to the energy history (reward_creator.py:63)
    assert seen == set(range(24))
