"""Value-level pins of the device-side reset.

1. Against the REFERENCE: the draws its Weather_Manager.reset made (noise array, roll, day, hour; fixture
   weather_resets.npz) are injected one level before the arithmetic (sdc_reset_override.noise / roll_days) and the
   device's windows, 30-day bounds and reset observation inputs must equal what the reference's manager held --
   roll direction and normalisation window included (one engine runs 30-day episodes, so the whole 2880-sample
   window is inside t_win).
2. Against a NumPy restatement of the device's own draw scheme (tests/reset_ref.py: Philox4x32-10, multiply-shift
   ranges; Philox4x32-7, fp32 Box-Muller, fp64 walk / std 0.75): day / hour / roll exactly, windows and bounds to 2e-6 C
   (the device's v_log_f32 / v_sin_f32 / v_cos_f32 are hardware approximations of the fp32 functions NumPy evaluates)."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
from tests import gpu_helpers as G
from tests import reset_ref as RR

pytestmark = pytest.mark.gpu
TL = L.TABLE_LEN


def _engine(n, steps, tables, seed=0, **kw):
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    eng = SdcEngine(n, episode_steps=steps, n_locations=len(tables), auto_reset=False, seed=seed, **kw)
    for i, tb in enumerate(tables):
        eng.set_tables(i, tb["W"], tb["C"], tb["T"], tb["WB"])
    eng.set_dc_params(0, p)
    return eng


@pytest.mark.parametrize("steps", [2880, 672])
def test_injected_reference_draws_give_the_reference_windows(steps):
    d = G.load_fixture("weather_resets")
    cases = [k for k in range(int(d["meta_cases"])) if int(d[f"case{k}_tz"]) == 0]
    syn = traces.synthetic_tables("ny", 0)
    locs = [str(x) for x in d["meta_locations"]]
    tables = [dict(W=syn["W"], C=syn["C"], T=d[loc + "_T"], WB=d[loc + "_WB"]) for loc in locs]
    N = len(cases)
    eng = _engine(N, steps, tables)
    loc_id = np.array([locs.index(str(d[f"case{k}_loc"])) for k in cases])
    eng.assign(loc_id, 0, 0, 364)
    noise = np.zeros((N, TL))
    roll = np.zeros(N, np.int32)
    for j, k in enumerate(cases):
        noise[j], roll[j] = RR.coherent_noise_legacy(int(d[f"case{k}_seed"]))
        np.testing.assert_allclose(noise[j][::32], d[f"case{k}_noise_sub32"], rtol=0, atol=1e-12)
        assert roll[j] == int(d[f"case{k}_roll_days"])
    day = np.array([int(d[f"case{k}_day"]) for k in cases], np.int32)
    hour = np.array([int(d[f"case{k}_hour"]) for k in cases], np.int32)
    obs, _ = eng.reset(override=dict(day=day, hour=hour, roll_days=roll, noise=noise))
    lw = eng.lw
    assert lw == steps + 18
    tw, wb = eng.get_state("t_win"), eng.get_state("wb_win")
    tmin, tden = eng.get_state("t_min"), eng.get_state("t_den")
    cur = eng.get_state("cursor")
    raw = G.raw_obs(obs.cpu().numpy())
    for j, k in enumerate(cases):
        pre = f"case{k}_"
        assert cur[j] == int(d[pre + "cursor0"])
        np.testing.assert_allclose(tw[j], d[pre + "T_win"][:lw], rtol=0, atol=1e-11)
        np.testing.assert_allclose(wb[j], d[pre + "WB_win"][:lw], rtol=0, atol=1e-11)
        assert abs(tmin[j] - float(d[pre + "t_min30"])) <= 1e-11
        assert abs(tmin[j] + tden[j] - float(d[pre + "t_max30"])) <= 1e-11
        if steps == 2880:      # the whole normalisation window is inside the episode window
            assert tmin[j] == tw[j][:2880].min() and tmin[j] + tden[j] == pytest.approx(tw[j][:2880].max(), abs=1e-12)
        # the observation's normalised temperatures (ls obs 14, dc obs 12 / 13) are the reference's NT[c0], NT[c0 + 1]
        nt = d[pre + "NT_win"]
        assert abs(raw[j, 14] - np.float32(nt[0])) <= 1e-6 and abs(raw[j, 26 + 13] - np.float32(nt[1])) <= 1e-6
    eng.close()


@pytest.mark.parametrize("steps,n_envs", [(672, 256), (2880, 24)])
def test_device_draws_match_the_numpy_restatement(steps, n_envs):
    tb = traces.synthetic_tables("ny", 0)
    seed, base = 0x1234ABCD5, 1000
    eng = _engine(n_envs, steps, [tb], seed=seed, env_index_base=base)
    months = np.arange(n_envs) % 12
    d0 = np.array([traces.get_init_day(int(m)) for m in months])
    lo, hi = np.maximum(0, d0 - 7), np.minimum(364, d0 + 7)
    eng.assign(0, 0, lo, hi)
    worst = 0.0
    for episode in (1, 2):
        eng.reset()
        day, hq, cur = eng.get_state("day"), eng.get_state("hourq"), eng.get_state("cursor")
        tw, wb = eng.get_state("t_win"), eng.get_state("wb_win")
        tmin, tden = eng.get_state("t_min"), eng.get_state("t_den")
        cmin, cden = eng.get_state("ci_min"), eng.get_state("ci_den")
        assert (eng.get_state("episode") == episode).all()
        check = range(n_envs) if n_envs <= 32 else list(range(0, n_envs, 7)) + [n_envs - 1]
        for i in check:
            x = RR.device_reset_expected(tb, seed, base + i, episode, int(lo[i]), int(hi[i]), steps)
            assert (day[i], hq[i] // 4, cur[i]) == (x["day"], x["hour"], x["c0"]), (i, episode)
            assert cmin[i] == x["ci_min"] and cden[i] == x["ci_den"]
            e = max(np.abs(tw[i] - x["t_win"]).max(), np.abs(wb[i] - x["wb_win"]).max(), abs(tmin[i] - x["t_min"]),
                    abs(tden[i] - x["t_den"]))
            worst = max(worst, float(e))
        # all envs: the draws (exact integer arithmetic)
        dd, hh, _ = RR.device_draws(seed, base + np.arange(n_envs), episode, lo, hi)
        fence = TL - 1 - (steps + 17)
        ok = dd * 96 + hh * 4 <= fence
        np.testing.assert_array_equal(day[ok], dd[ok])
        np.testing.assert_array_equal(hq[ok] // 4, hh[ok])
    print("device reset vs NumPy restatement: max |dT| =", worst)
    assert worst <= 2e-6      # measured 8.4e-7
    eng.close()


def test_device_reset_without_weather_noise_is_exact():
    """`weather_noise_std = 0`: the device draws day / hour / roll, and the windows are the table rolled and clipped -- no
    transcendental anywhere, so every value equals the NumPy restatement's exactly (the kernel's other branch: the walk and
    the second visit of its window blocks are skipped, the batched add / roll / clip pass runs instead)."""
    tb = traces.synthetic_tables("ny", 0)
    seed, base, steps, n = 77, 5000, 672, 96
    eng = _engine(n, steps, [tb], seed=seed, env_index_base=base, weather_noise_std=0.0)
    lo, hi = np.full(n, 3), np.full(n, 361)          # the whole year: cursors near both ends, rolls that wrap
    eng.assign(0, 0, lo, hi)
    eng.reset()
    day, hq, cur = eng.get_state("day"), eng.get_state("hourq"), eng.get_state("cursor")
    tw, wb = eng.get_state("t_win"), eng.get_state("wb_win")
    tmin, tden = eng.get_state("t_min"), eng.get_state("t_den")
    for i in range(n):
        x = RR.device_reset_expected(tb, seed, base + i, 1, int(lo[i]), int(hi[i]), steps, noise_std=0.0)
        assert (day[i], hq[i] // 4, cur[i]) == (x["day"], x["hour"], x["c0"]), i
        assert np.array_equal(tw[i], x["t_win"]) and np.array_equal(wb[i], x["wb_win"]), i
        assert tmin[i] == x["t_min"] and tden[i] == x["t_den"], i
    eng.close()
