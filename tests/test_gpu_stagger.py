"""Environments that are NOT in lock-step (VERDICT r1 "untested hot-path branch"): masked sdc_reset in the middle of an
episode, mixed episode steps inside one launch (the host passes rel_hint = -1, the kernel reads each env's own
feature row), episodes that end at different steps, auto-reset of a SUBSET of the batch, and actions outside
{0,1,2} -- all compared step by step with the fp64 CPU oracle (1e-5, verify mode on)."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from tests import gpu_helpers as G
from tests import parity_util as P
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check_reset_obs(eobs, oobs, worst):
    for i, o in oobs.items():
        worst["obs"] = max(worst["obs"], float(G.rel_err(eobs[i], o).max()))


def test_masked_reset_mid_episode_and_staggered_terminals_vs_oracle():
    N, steps = 48, 96
    rig = P.ParityRig(N, episode_steps=steps, seed=71, dc_files=("dc_config.json", "dc_config_r16.json"))
    arng = np.random.default_rng(72)
    worst = dict(obs=0.0, rew=0.0, info=0.0)
    eobs, oobs = rig.reset_all()
    _check_reset_obs(eobs, oobs, worst)
    t_rel = np.zeros(N, dtype=int)
    n_masked = n_mixed_steps = 0
    for t in range(260):
        acts = arng.integers(0, 3, size=(N, 3)).astype(np.int32)
        ed = P.compare_step(rig, acts, worst)
        t_rel += 1
        np.testing.assert_array_equal(ed.astype(bool), t_rel >= steps)
        if len(np.unique(t_rel)) > 1:
            n_mixed_steps += 1
        # masked resets: the finished envs (must be reset before the next step: auto_reset is off), and on three
        # occasions a third / a half / one env of the batch in the MIDDLE of their episodes
        mask = t_rel >= steps
        if t == 30:
            mask = mask | (np.arange(N) % 3 == 0)
        if t == 55:
            mask = mask | (np.arange(N) >= N // 2)
        if t == 57:
            mask = mask | (np.arange(N) == 5)
        if mask.any():
            eobs, oobs = rig.reset_some(mask)
            _check_reset_obs(eobs, oobs, worst)
            assert set(oobs) == set(np.nonzero(mask)[0].tolist())
            t_rel[mask] = 0
            n_masked += 1
            np.testing.assert_array_equal(rig.eng.get_state("t_rel"), t_rel)
            assert rig.eng.steps_to_episode_end() == steps - t_rel.max()
    print("staggered:", worst, "masked resets", n_masked, "steps with mixed episode positions", n_mixed_steps)
    assert n_masked >= 6 and n_mixed_steps >= 200
    assert worst["obs"] <= TOL and worst["rew"] <= TOL and worst["info"] <= 2e-6
    assert (rig.eng.get_state("order_stat_sticky") == 0).all()
    rig.eng.close()


def test_auto_reset_of_a_subset_vs_oracle():
    """auto_reset on, envs staggered by a masked reset: when only SOME envs finish, sdc_step resets exactly those on
    the device (Philox draws), returns their reset obs and their pre-reset obs in final_obs, and leaves the others
    alone.  The oracle restarts a finished env from the windows the device drew (read back), so every later step of
    every env is still checked."""
    import torch
    from dc_rl_amd import dc_config, traces
    from dc_rl_amd.engine import SdcEngine
    N, steps = 40, 64
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    eng = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=19, debug_flags=1)
    eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
    eng.set_dc_params(0, p)
    init_day = traces.get_init_day(4)
    eng.assign(0, 0, init_day - 7, init_day + 7)
    op = G.oracle_params_from_dict(p)
    orcs = [po.OracleEnv(op) for _ in range(N)]
    for o in orcs:
        o.e.stpt = float(p["init_setpoint"])

    def oracle_begin_from_device(envs, raw_obs):
        day, hq, cur = eng.get_state("day"), eng.get_state("hourq"), eng.get_state("cursor")
        tw, wb = eng.get_state("t_win"), eng.get_state("wb_win")
        tmin, tden = eng.get_state("t_min"), eng.get_state("t_den")
        cmin, cden = eng.get_state("ci_min"), eng.get_state("ci_den")
        for i in envs:
            c0 = int(cur[i])
            lo, hi = max(0, c0 - 16), c0 + steps + 18
            T, WBv = np.zeros(hi - lo), np.zeros(hi - lo)
            T[c0 - lo:] = tw[i]
            WBv[c0 - lo:] = wb[i]
            NC = (tb["C"][lo:hi] - cmin[i]) / cden[i]
            NT = (T - tmin[i]) / tden[i]
            oo = orcs[i].begin(tb["W"][lo:hi], tb["C"][lo:hi], NC, T, WBv, NT, lo, int(day[i]), int(hq[i]) // 4, steps)
            assert G.rel_err(raw_obs[i], oo).max() <= TOL, i

    obs, share = eng.reset()
    oracle_begin_from_device(range(N), G.raw_obs(obs.cpu().numpy()))
    t_rel = np.zeros(N, dtype=int)
    rng = np.random.default_rng(20)
    worst = dict(obs=0.0, rew=0.0)
    subset_resets = 0
    for t in range(200):
        if t == 20:      # stagger: a third of the envs restart now (device draws), so episodes end at two different steps
            m = (np.arange(N) % 3 == 1)
            obs, share = eng.reset(mask=m.astype(np.uint8))
            oracle_begin_from_device(np.nonzero(m)[0], G.raw_obs(obs.cpu().numpy()))
            t_rel[m] = 0
        acts = rng.integers(0, 3, size=(N, 3)).astype(np.int32)
        obs, share, rew, done, info = eng.step(torch.from_numpy(acts).to(eng.device))
        eo, er, ed = G.raw_obs(obs.cpu().numpy()), rew.cpu().numpy(), done.cpu().numpy().astype(bool)
        fo = G.raw_obs(eng.final_obs.cpu().numpy())
        t_rel += 1
        np.testing.assert_array_equal(ed, t_rel >= steps)
        for i in range(N):
            oo, orew, odone, oinfo = orcs[i].step(acts[i])
            assert bool(odone) == bool(ed[i])
            got = fo[i] if ed[i] else eo[i]      # a finished env's last observation is in final_obs; obs holds the reset obs
            worst["obs"] = max(worst["obs"], float(G.rel_err(got, oo).max()))
            worst["rew"] = max(worst["rew"], float(G.rel_err(er[i], orew).max()))
        if ed.any():
            assert not ed.all()
            subset_resets += 1
            fin = np.nonzero(ed)[0]
            assert (eng.get_state("t_rel")[fin] == 0).all() and (eng.get_state("t_rel")[~ed] == t_rel[~ed]).all()
            oracle_begin_from_device(fin, eo)   # (checks the returned reset obs against the oracle's)
            t_rel[ed] = 0
    print("subset auto-reset:", worst, "boundaries", subset_resets)
    assert subset_resets >= 5
    assert worst["obs"] <= TOL and worst["rew"] <= TOL
    assert (eng.get_state("order_stat_sticky") == 0).all() and (eng.get_state("fault") == 0).all()
    eng.close()


def test_out_of_range_actions_set_the_fault_bit():
    """agent_dc / agent_bat actions outside {0,1,2} raise KeyError in the reference (envs/dc_gym.py:160,
    bat_env_fwd_view.py:99); the device flags SDC_FAULT_ACTION (sticky for the episode) and steps the env as if the
    action were "no change" / "idle" -- the other envs of the batch are untouched."""
    N, steps = 8, 96
    rig = P.ParityRig(N, episode_steps=steps, seed=5)
    rig.reset_all()
    arng = np.random.default_rng(6)
    FAULT_ACTION = 64
    for t in range(6):
        acts = arng.integers(0, 3, size=(N, 3)).astype(np.int32)
        sent = acts.copy()
        if t == 2:
            sent[1, 1] = 7      # dc
            acts[1, 1] = 1      # ... steps as "no change"
            sent[3, 2] = -1     # bat
            acts[3, 2] = 2      # ... steps as "idle"
            sent[6, 0] = 3      # ls: "do nothing" in the reference too (carbon_ls.py:266)
            acts[6, 0] = 1
        eo, es, er, ed, ei = rig.step(sent)
        f = ei[:, L.INFO_IDX["fault"]].astype(int)
        bad = np.zeros(N, dtype=bool)
        if t >= 2:
            bad[[1, 3, 6]] = True
        np.testing.assert_array_equal((f & FAULT_ACTION) != 0, bad)
        assert ((f & ~FAULT_ACTION) == 0).all()
        for i, orc in rig.oracles.items():
            oo, orew, odone, oinfo = orc.step(acts[i])
            assert G.rel_err(eo[i], oo).max() <= TOL and G.rel_err(er[i], orew).max() <= TOL
    rig.eng.close()


def test_odd_batch_and_more_than_32_racks_vs_oracle(tmp_path):
    """Two corners of the two-envs-per-wavefront kernel: an ODD number of envs (the last wavefront carries one env; its
    other half mirrors it and stores nothing) and a data centre with MORE RACKS THAN A HALF-WAVE HAS LANES (40 racks: the
    rack model takes a second pass), mixed with the default 20-rack config in one batch."""
    import json
    import os
    from dc_rl_amd import dc_config
    src = os.path.join(os.path.dirname(dc_config.__file__), "configs", "dc_config.json")
    cfg = json.load(open(src))
    d = cfg["data_center_configuration"]
    d["NUM_ROWS"], d["NUM_RACKS_PER_ROW"] = 8, 5
    d["RACK_SUPPLY_APPROACH_TEMP_LIST"] = (d["RACK_SUPPLY_APPROACH_TEMP_LIST"] * 2)[:40]
    d["RACK_RETURN_APPROACH_TEMP_LIST"] = (d["RACK_RETURN_APPROACH_TEMP_LIST"] * 2)[:40]
    sv = cfg["server_characteristics"]
    sv["DEFAULT_SERVER_POWER_CHARACTERISTICS"] = (sv["DEFAULT_SERVER_POWER_CHARACTERISTICS"] * 2)[:40]
    big = str(tmp_path / "dc_config_r40.json")
    json.dump(cfg, open(big, "w"))
    N, steps = 7, 96
    rig = P.ParityRig(N, episode_steps=steps, seed=41, dc_files=("dc_config.json", big))
    assert sorted({len(p["rack_n"]) for p in rig.params}) == [20, 40]
    arng = np.random.default_rng(42)
    worst = dict(obs=0.0, rew=0.0, info=0.0)
    eobs, oobs = rig.reset_all()
    _check_reset_obs(eobs, oobs, worst)
    for t in range(steps):
        P.compare_step(rig, arng.integers(0, 3, size=(N, 3)).astype(np.int32), worst)
    print("odd batch, 20 + 40 racks:", worst)
    assert worst["obs"] <= TOL and worst["rew"] <= TOL and worst["info"] <= 2e-6
    rig.eng.close()
