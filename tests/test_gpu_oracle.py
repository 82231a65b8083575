"""HIP path vs the fp64 CPU oracle on identical injected inputs, plus the
device-side reset and size-independent properties at the full 4096-env size."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
from oracle import pyoracle as po
from tests import gpu_helpers as G
from tests import parity_util as P

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star: 1e-5 relative fp32 (absolute where |ref| < 1)


def test_config2_256_envs_two_episodes_vs_oracle():
    """256 envs, one DC config, months cycling rank % 12, 2 episodes (auto-reset boundary crossed)."""
    w = P.run_engine_vs_oracle(n_envs=256, n_steps=2 * 672, episode_steps=672, seed=2, locations=("ny",))
    print("config2", w)
    assert w["obs"] <= TOL and w["rew"] <= TOL and w["info"] <= TOL


def test_config4_heterogeneous_rack_counts_vs_oracle():
    """16 / 20 / 25-rack configs interleaved by env id, three locations."""
    w = P.run_engine_vs_oracle(n_envs=96, n_steps=400, episode_steps=288, seed=4, locations=("ny", "az", "wa"),
                               dc_files=("dc_config.json", "dc_config_r16.json", "dc_config_r25.json"))
    print("config4", w)
    assert w["obs"] <= TOL and w["rew"] <= TOL and w["info"] <= TOL


def test_full_history_ring_vs_oracle():
    """History rings pre-filled to 10 000 entries (steady state): order statistics + clipped mean/std vs oracle."""
    import torch
    N = 32
    rig = P.ParityRig(N, episode_steps=96, seed=7)
    rng = np.random.default_rng(7)
    hist = np.full((N, rig.eng.hist_stride), np.nan, np.float32)   # NaN = empty slot
    vals = (331 + 70 * rng.standard_normal((N, 10000))).clip(150, 650).astype(np.float32)
    vals[:, ::97] = vals[:, 5:6]          # duplicates on purpose
    hist[:, :10000] = vals
    pos = rng.integers(0, 10000, N).astype(np.int32)
    rig.eng.set_state("hist", hist)
    rig.eng.set_state("hist_len", np.full(N, 10000, np.int32))
    rig.eng.set_state("hist_pos", pos)
    for i, orc in rig.oracles.items():
        orc.e.hist_len = 10000
        orc.e.hist_pos = int(pos[i])
        np.ctypeslib.as_array(orc.e.hist)[:] = vals[i].astype(np.float64)
    worst = dict(obs=0.0, rew=0.0, info=0.0)
    rig.reset_all()
    arng = np.random.default_rng(8)
    for t in range(96):
        P.compare_step(rig, arng.integers(0, 3, (N, 3)).astype(np.int32), worst)
    print("full ring", worst)
    assert worst["rew"] <= TOL and worst["obs"] <= TOL
    rig.eng.close()


def test_order_statistic_tracker_stress():
    """The O(1) order-statistic trackers of the reward normalisation against the exact bisection (debug_flags bit 0) on
    histories built to stress them: heavy duplicates, monotone drifts, constant runs, values straddling zero (sign
    change of the fp32 offsets), a small history capacity so evictions start early."""
    import torch
    N, steps, cap = 64, 96, 257
    rig = P.ParityRig(N, episode_steps=steps, seed=21, hist_cap=cap, with_oracle=False)
    eng = rig.eng
    rng = np.random.default_rng(21)
    hist = np.full((N, eng.hist_stride), np.nan, np.float32)
    L0 = 200
    base = rng.standard_normal((N, L0)).astype(np.float32) * 40
    base[0::4] = np.round(base[0::4] / 20) * 20            # few distinct values
    base[1::4] = np.sort(base[1::4], axis=1)               # monotone
    base[2::4, 50:150] = 7.0                               # long constant run
    hist[:, :L0] = base
    eng.set_state("hist", hist)
    eng.set_state("hist_len", np.full(N, L0, np.int32))
    eng.set_state("hist_pos", np.zeros(N, np.int32))
    rig.reset_all()
    paths = np.zeros(4, np.int64)
    for ep in range(6):
        for t in range(steps):
            a = torch.randint(0, 3, (N, 3), dtype=torch.int32, device=eng.device)
            if ep % 2 == 1:
                a[:, :] = 1
                a[:, 2] = 2                                # idle policy: near-constant energies -> duplicates
            obs, share, rew, done, info = eng.step(a)
            inf = info.cpu().numpy()
            assert (inf[:, L.INFO_IDX["fault"]] == 0).all(), (ep, t, inf[:, L.INFO_IDX["fault"]])
            paths += np.bincount(inf[:, 39].astype(int), minlength=4)[:4]
        rig.reset_all()
    assert (eng.get_state("order_stat_sticky") == 0).all()
    assert (eng.get_state("hist_len") == cap).all()
    print("paths (windows only, a window re-centred ahead of need, -, rebuild from the ring):", paths)
    assert paths[0] > 3 * paths[1]                         # the window usually answers without a sweep
    eng.close()


def test_clip_bound_windows_heavy_tails_and_drifting_bounds():
    """The running tail sums and clip-bound windows of the reward normalisation (verify mode checks every step
    against a direct pass over the ring): histories with 0.5 % .. 9 % of the keys beyond a clip bound -- a tail is
    only a count and two sums, however heavy -- and a drifting level that keeps the bounds moving."""
    import torch
    N, steps, cap = 64, 200, 10000
    rig = P.ParityRig(N, episode_steps=steps, seed=33, hist_cap=cap, with_oracle=False)
    eng = rig.eng
    rng = np.random.default_rng(33)
    hist = np.full((N, eng.hist_stride), np.nan, np.float32)
    frac = np.array([0.005, 0.02, 0.04, 0.09])[np.arange(N) % 4]
    base = rng.standard_normal((N, cap)) * 20
    out = rng.random((N, cap)) < frac[:, None]
    base = np.where(out, base + np.sign(rng.standard_normal((N, cap))) * (120 + 60 * rng.random((N, cap))), base)
    base += np.linspace(0, 1, cap)[None, :] * (np.arange(N)[:, None] % 3 - 1) * 30     # drift: none / up / down
    hist[:, :cap] = base.astype(np.float32)
    eng.set_state("hist", hist)
    eng.set_state("hist_len", np.full(N, cap, np.int32))
    eng.set_state("hist_pos", np.zeros(N, np.int32))
    rig.reset_all()
    paths = np.zeros(4, np.int64)
    for t in range(steps):
        a = torch.randint(0, 3, (N, 3), dtype=torch.int32, device=eng.device)
        obs, share, rew, done, info = eng.step(a)
        inf = info.cpu().numpy()
        assert (inf[:, L.INFO_IDX["fault"]] == 0).all(), (t, np.nonzero(inf[:, L.INFO_IDX["fault"]])[0])
        assert np.isfinite(rew.cpu().numpy()).all()
        paths += np.bincount(inf[:, 39].astype(int), minlength=4)[:4]
    assert (eng.get_state("order_stat_sticky") == 0).all()
    print("paths (no ring read, a window re-centred inline, a deferred re-centred window taken over, rebuilt):", paths)
    assert paths[3] <= 2 * N                # one rebuild each after the injection, not one per step -- heavy tails included
    assert paths[0] + paths[2] > 20 * (paths[1] + paths[3])   # the incremental state serves nearly every step
    eng.close()


def test_device_reset_and_auto_reset():
    """Device-side reset (Philox draws, coherent noise, roll, clip, 30-day min/max): distributional checks and
    the reset observation recomputed by the oracle from the windows the device produced."""
    import torch
    N, steps = 512, 96
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    eng = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=11)
    eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
    eng.set_dc_params(0, p)
    init_day = traces.get_init_day(6)
    eng.assign(0, 0, init_day - 7, init_day + 7)
    obs, share = eng.reset()
    raw0 = G.raw_obs(obs.cpu().numpy())
    day, hq, cur = eng.get_state("day"), eng.get_state("hourq"), eng.get_state("cursor")
    assert day.min() >= init_day - 7 and day.max() <= init_day + 7 and len(np.unique(day)) == 15
    assert (hq % 4 == 0).all() and hq.min() == 0 and hq.max() == 92 and (cur == day * 96 + hq).all()
    tw, wb = eng.get_state("t_win"), eng.get_state("wb_win")
    assert tw.min() >= 0 and tw.max() <= 45 and wb.min() >= 0 and wb.max() <= 45
    tmin, tden = eng.get_state("t_min"), eng.get_state("t_den")
    cmin, cden = eng.get_state("ci_min"), eng.get_state("ci_den")
    for i in range(0, N, 37):
        c0 = int(cur[i])
        assert cmin[i] == tb["C"][c0:c0 + 2880].min() and np.isclose(cden[i], np.ptp(tb["C"][c0:c0 + 2880]), rtol=0, atol=0)
        assert tmin[i] <= tw[i].min() + 1e-12 and tmin[i] + tden[i] >= tw[i].max() - 1e-12
    # the noise is a random walk rescaled to std 0.75 over the YEAR; over a 1-day window its deviation from the
    # un-noised table must be smooth (small increments) and different between envs
    # oracle recomputes the reset obs from the device-produced windows
    for i in range(0, N, 61):
        c0 = int(cur[i])
        lo, hi = max(0, c0 - 16), c0 + steps + 18
        T = np.zeros(hi - lo)
        WBv = np.zeros(hi - lo)
        T[c0 - lo:] = tw[i]
        WBv[c0 - lo:] = wb[i]
        NC = (tb["C"][lo:hi] - cmin[i]) / cden[i]
        NT = (T - tmin[i]) / tden[i]
        orc = po.OracleEnv(G.oracle_params_from_dict(p))
        oo = orc.begin(tb["W"][lo:hi], tb["C"][lo:hi], NC, T, WBv, NT, lo, int(day[i]), int(hq[i]) // 4, steps)
        assert G.rel_err(raw0[i], oo).max() <= TOL
    # auto-reset: after `steps` steps every env is done, obs are reset obs, final_obs holds the last obs
    acts = torch.randint(0, 3, (N, 3), dtype=torch.int32, device=eng.device)
    for t in range(steps):
        obs, share, rew, done, info = eng.step(acts)
        if t < steps - 1:
            assert int(done.sum()) == 0
    assert int(done.sum()) == N
    assert (eng.get_state("t_rel") == 0).all() and (eng.get_state("episode") == 2).all()
    assert (eng.get_state("bat_load") == 0).all() and (eng.get_state("q_cum") == 0).all()
    assert (eng.get_state("hist_len") == steps).all()           # history survives reset
    cur2 = eng.get_state("cursor")
    assert (cur2 != cur + steps).any()                           # a new start was drawn
    assert not np.array_equal(eng.get_state("t_win"), tw)        # new noise realisation
    fo, o2 = G.raw_obs(eng.final_obs.cpu().numpy()), G.raw_obs(obs.cpu().numpy())
    assert np.abs(fo[:, -1]).max() >= 0 and not np.array_equal(fo, o2)
    # next step works without an explicit reset
    eng.step(acts)
    eng.close()


def test_weather_noise_statistics():
    """Coherent noise: with zero base tables the device window IS clip(noise); its year-std target is 0.75
    (checked loosely through the window) and increments follow the 0.02-weighted walk scaling."""
    N, steps = 256, 2880
    z = np.zeros(L.TABLE_LEN)
    base = np.full(L.TABLE_LEN, 20.0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    eng = SdcEngine(N, episode_steps=steps, auto_reset=False, seed=5)
    tb = traces.synthetic_tables("ny", 0)
    eng.set_tables(0, tb["W"], tb["C"], base, base)
    eng.set_dc_params(0, p)
    eng.assign(0, 0, 100, 100)
    eng.reset()
    tw = eng.get_state("t_win") - 20.0
    inc = np.diff(tw, axis=1)
    # increments = 0.02 * N(0,1) * (0.75 / std_walk); std_walk of a 35040-step walk ~ 0.02*sqrt(35040)*O(0.3..0.8)
    assert np.isfinite(tw).all()
    assert 0.001 < inc.std() < 0.02
    assert abs(inc.mean()) < 1e-3
    # different envs get different realisations; wet bulb shares the SAME noise (managers.py:598-599)
    assert np.abs(tw[0] - tw[1]).max() > 1e-3
    np.testing.assert_allclose(eng.get_state("wb_win"), eng.get_state("t_win"), rtol=0, atol=1e-12)
    # year-scale std across envs: each env's window sample has |value| typically within a few x 0.75
    assert np.abs(tw).max() < 6.0
    eng.close()


def test_full_size_properties_4096():
    """BASELINE config 3 size: determinism, independence of an env from its batch, state checkpoint round trip."""
    import torch
    N, steps = 4096, 672
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)

    def mk(n, seed=3):
        e = SdcEngine(n, episode_steps=steps, auto_reset=True, seed=seed)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 174, 188)
        return e

    g = torch.Generator(device="cpu").manual_seed(0)
    acts = torch.randint(0, 3, (50, N, 3), dtype=torch.int32, generator=g).cuda()
    a, b, small = mk(N), mk(N), mk(64)
    oa, _ = a.reset()
    ob, _ = b.reset()
    os_, _ = small.reset()
    assert torch.equal(oa, ob) and torch.equal(oa[:64], os_)
    ra = []
    for t in range(50):
        xa = [x.clone() for x in a.step(acts[t])]
        xb = b.step(acts[t])
        xs = small.step(acts[t, :64].contiguous())
        for u, v in zip(xa, xb):
            assert torch.equal(u, v)                      # bitwise deterministic
        for u, v in zip(xa, xs):
            assert torch.equal(u[:64], v)                 # an env does not depend on its batch
        ra.append(xa[2])
    assert torch.isfinite(torch.stack(ra)).all()
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    # checkpoint round trip: load a's state into a fresh engine and continue identically
    c = mk(N, seed=3)
    c.reset()
    c.load_state_dict(a.state_dict())
    g2 = torch.randint(0, 3, (N, 3), dtype=torch.int32, device="cuda")
    ya = [x.clone() for x in a.step(g2)]
    yc = c.step(g2)
    for u, v in zip(ya, yc):
        assert torch.equal(u, v)
    for e in (a, b, small, c):
        e.close()


def test_long_episode_without_feature_rows_vs_oracle():
    """Episodes too long for the per-episode feature rows (the features kernel keeps an episode's windows in LDS): the
    step computes the trace-only observation entries itself, and the result still matches the oracle."""
    w = P.run_engine_vs_oracle(n_envs=16, n_steps=150, episode_steps=4000, seed=17)
    print("long episode", w)
    assert w["obs"] <= TOL and w["rew"] <= TOL and w["info"] <= 2e-6


def test_grid_shapes_with_sweep_workgroups_inside_the_grid_vs_oracle():
    """Batches whose step grid is larger than the point where the 32 spare (sweep) workgroups are inserted (320 pair
    workgroups = 2560 envs) and NOT a multiple of 8 workgroups (the XCD-contiguous env mapping falls back to the
    identity), with an odd number of envs: the envs around the insertion point, the first and the last ones against the
    oracle across an episode boundary."""
    for n in (2603, 2570):
        last = n - 1
        envs = sorted({0, 1, 2, 3, 1279, 1280, 2551, 2552, 2559, 2560, 2561, 2567, 2568, last - 2, last - 1, last})
        w = P.run_engine_vs_oracle(n_envs=n, n_steps=130, episode_steps=96, seed=31 + n, oracle_envs=envs)
        print(n, w)
        assert w["obs"] <= 1e-5 and w["rew"] <= 1e-5 and w["info"] <= 2e-6


def test_checkpoint_rollback_on_the_same_engine_4096():
    """state_dict(), one step, load_state_dict() on the SAME engine, the same step again: bit-identical outputs, verify
    mode on (debug_flags bit 0), 4096 envs with full rings so that deferred window re-centrings are in flight at every
    checkpoint (a restored header must not take over a window swept for the state it replaced: sdc_set_state moves the
    launch counter past every request stamp)."""
    import torch
    N, steps, cap = 4096, 672, 10000
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    eng = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=5, debug_flags=1)
    eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
    eng.set_dc_params(0, p)
    eng.assign(0, 0, 174, 188)
    rng = np.random.default_rng(5)
    hist = np.full((N, eng.hist_stride), np.nan, np.float32)
    hist[:, :cap] = (331 + 70 * rng.standard_normal((N, cap))).clip(150, 650).astype(np.float32)
    eng.set_state("hist", hist)
    eng.set_state("hist_len", np.full(N, cap, np.int32))
    eng.set_state("hist_pos", rng.integers(0, cap, N).astype(np.int32))
    del hist
    eng.reset()
    g = torch.Generator(device="cpu").manual_seed(6)
    taken = 0
    for t in range(40):
        eng.step(torch.randint(0, 3, (N, 3), dtype=torch.int32, generator=g).cuda())
    for rnd in range(3):
        sd = eng.state_dict()
        pend = int((sd["header"][:, _hdr_pend()] != 0).any(axis=1).sum())
        a1 = torch.randint(0, 3, (N, 3), dtype=torch.int32, generator=g).cuda()
        a2 = torch.randint(0, 3, (N, 3), dtype=torch.int32, generator=g).cuda()
        first = [[x.clone() for x in eng.step(a1)], [x.clone() for x in eng.step(a2)]]
        eng.load_state_dict(sd)
        again = [[x.clone() for x in eng.step(a1)], [x.clone() for x in eng.step(a2)]]
        for k in range(2):
            for u, v, nm in zip(first[k], again[k], ("obs", "share_obs", "rew", "done", "info")):
                if nm == "info":     # (the diagnostics column says HOW the reward state was served, which a restore may change)
                    u, v = u.clone(), v.clone()
                    u[:, L.INFO_IDX["reserved"]] = 0
                    v[:, L.INFO_IDX["reserved"]] = 0
                assert torch.equal(u, v), (rnd, k, nm)
        assert (eng.info[:, L.INFO_IDX["fault"]] == 0).all()
        taken += pend
        for t in range(7):
            eng.step(torch.randint(0, 3, (N, 3), dtype=torch.int32, generator=g).cuda())
            assert (eng.info[:, L.INFO_IDX["fault"]] == 0).all()
    assert (eng.get_state("order_stat_sticky") == 0).all()
    print("envs with a re-centring request in flight at the checkpoints:", taken)
    assert taken > 0
    eng.close()


def _hdr_pend():
    """dword indices of the four in-flight re-centring stamps in the 256-byte header (csrc/sdc_device.hpp H_PEND)."""
    import re, os
    src = open(os.path.join(os.path.dirname(L.LIB_PATH), "sdc_device.hpp")).read()
    m = re.search(r"H_PEND\s*=\s*(\d+)", src)
    assert m, "H_PEND not found in sdc_device.hpp"
    k = int(m.group(1))
    return [k, k + 1, k + 2, k + 3]
