"""Engine-vs-oracle parity driver used by the -m gpu tests, smoke() and bench.py's self-check.

Inputs are synthetic (seeded): trace tables from dc_rl_amd.traces.synthetic_tables, per-env start day / hour,
0..13-day roll and coherent weather noise drawn with NumPy on the host exactly the way the reference's
reset does (sustaindc_env.py:454-455, utils/managers.py:35-48, :596-613), then INJECTED into both the HIP
engine (sdc_reset override) and the fp64 CPU oracle, so both see identical trace inputs.
"""
from __future__ import annotations

import numpy as np

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
from oracle import pyoracle as po
from tests import gpu_helpers as G
from tests.reset_ref import reference_weather_reset

TL = L.TABLE_LEN


def host_reset_draw(rng, tables, day_lo, day_hi, steps, noise_std=0.75, noise_weight=0.02):
    """One env's reset draws + the derived per-episode arrays (full-year T / WB after noise, roll, clip)."""
    day = int(rng.integers(day_lo, day_hi + 1))
    hour = int(rng.integers(0, 24))
    c0 = day * 96 + hour * 4
    last_ok = TL - 1 - (steps + 17)
    if c0 > last_ok:  # year-end fence (same rule as csrc/sdc_reset.hip)
        day, hour = last_ok // 96, (last_ok % 96) // 4
        c0 = day * 96 + hour * 4
    walk = np.cumsum(noise_weight * rng.normal(0.0, 1.0, TL))
    noise = (walk / np.std(walk)) * noise_std
    roll = int(rng.integers(0, 14))
    # the arithmetic of Weather_Manager.reset, pinned against the reference by tests/golden/weather_resets.npz
    r = reference_weather_reset(tables["T"], tables["WB"], noise, roll, c0, TL)
    T, WB, tmin, tmax = r["T_full"], r["WB_full"], r["t_min"], r["t_max"]
    C = tables["C"]
    cmin, cmax = C[c0:c0 + 2880].min(), C[c0:c0 + 2880].max()
    return dict(day=day, hour=hour, c0=c0, T=T, WB=WB, t_min=tmin, t_max=tmax, ci_min=cmin, ci_max=cmax)


class ParityRig:
    """N envs on the engine + N scalar oracle envs fed the same inputs."""

    def __init__(self, n_envs, episode_steps=672, seed=0, locations=("ny",), dc_files=("dc_config.json",),
                 capacity_mw=1.0, months=None, hist_cap=10000, with_oracle=True, oracle_envs=None, debug_flags=1,
                 reward_method=(0, 0, 0)):
        self.N = n_envs
        self.steps = episode_steps
        self.rng = np.random.default_rng(seed)
        self.tables = [traces.synthetic_tables(loc, seed=seed) for loc in locations]
        combos = [(li, f) for li in range(len(locations)) for f in dc_files]
        self.params = []
        for li, f in combos:
            ci_loc, _ = traces.obtain_paths(locations[li])
            self.params.append(dc_config.size_datacenter(f, capacity_mw, traces.max_ambient_for_sizing(ci_loc)))
        self.eng = SdcEngine(n_envs, episode_steps=episode_steps, auto_reset=False, n_locations=len(locations),
                             n_dc_configs=len(combos), seed=seed, hist_cap=hist_cap, debug_flags=debug_flags,
                             reward_method=tuple(reward_method))
        for li, tb in enumerate(self.tables):
            self.eng.set_tables(li, tb["W"], tb["C"], tb["T"], tb["WB"])
        for ci, p in enumerate(self.params):
            self.eng.set_dc_params(ci, p)
        e = np.arange(n_envs)
        self.loc_id = (e % len(locations)).astype(np.int32)
        self.cfg_id = np.array([combos.index((int(self.loc_id[i]), dc_files[(i // len(locations)) % len(dc_files)]))
                                for i in range(n_envs)], dtype=np.int32)
        month = (e % 12) if months is None else np.asarray(months)
        init_day = np.array([traces.get_init_day(int(m)) for m in month])
        self.day_lo = np.maximum(0, init_day - 7).astype(np.int32)   # sustaindc_env.py:198
        self.day_hi = np.minimum(364, init_day + 7).astype(np.int32)
        self.eng.assign(self.loc_id, self.cfg_id, self.day_lo, self.day_hi)
        self.oracle_envs = list(range(n_envs)) if oracle_envs is None else list(oracle_envs)
        self.oracles = {}
        if with_oracle:
            for i in self.oracle_envs:
                self.oracles[i] = po.OracleEnv(G.oracle_params_from_dict(dict(self.params[self.cfg_id[i]],
                                                                              reward_method=tuple(reward_method))))
                self.oracles[i].e.stpt = float(self.params[self.cfg_id[i]]["init_setpoint"])

    def reset_all(self):
        """Inject a fresh episode everywhere.  Returns (engine raw obs [N,53], oracle raw obs {env: [53]})."""
        N, lw = self.N, self.eng.lw
        ov = dict(day=np.zeros(N, np.int32), hour=np.zeros(N, np.int32), ci_min=np.zeros(N), ci_max=np.zeros(N),
                  t_min=np.zeros(N), t_max=np.zeros(N), t_win=np.zeros((N, lw)), wb_win=np.zeros((N, lw)))
        oobs = {}
        for i in range(N):
            tb = self.tables[self.loc_id[i]]
            dr = host_reset_draw(self.rng, tb, self.day_lo[i], self.day_hi[i], self.steps)
            c0 = dr["c0"]
            for k in ("day", "hour", "ci_min", "ci_max", "t_min", "t_max"):
                ov[k][i] = dr[k]
            ov["t_win"][i] = dr["T"][c0:c0 + lw]
            ov["wb_win"][i] = dr["WB"][c0:c0 + lw]
            if i in self.oracles:
                lo, hi = max(0, c0 - 16), c0 + self.steps + 18
                NC = (tb["C"][lo:hi] - dr["ci_min"]) / (dr["ci_max"] - dr["ci_min"])
                NT = (dr["T"][lo:hi] - dr["t_min"]) / (dr["t_max"] - dr["t_min"])
                oobs[i] = self.oracles[i].begin(tb["W"][lo:hi], tb["C"][lo:hi], NC, dr["T"][lo:hi], dr["WB"][lo:hi], NT,
                                                lo, dr["day"], dr["hour"], self.steps)
        obs, _ = self.eng.reset(override=ov)
        return G.raw_obs(obs.cpu().numpy()), oobs

    def reset_some(self, mask):
        """Inject a fresh episode into the masked envs only (sdc_reset with a mask): the others keep stepping where
        they are, so the batch is no longer in lock-step.  Returns (engine raw obs [N,53], oracle raw obs {env: [53]})
        -- rows of unmasked envs are whatever the engine's obs buffer held."""
        N, lw = self.N, self.eng.lw
        mask = np.asarray(mask, dtype=bool)
        ov = dict(day=np.zeros(N, np.int32), hour=np.zeros(N, np.int32), ci_min=np.zeros(N), ci_max=np.ones(N),
                  t_min=np.zeros(N), t_max=np.ones(N), t_win=np.zeros((N, lw)), wb_win=np.zeros((N, lw)))
        oobs = {}
        for i in np.nonzero(mask)[0]:
            tb = self.tables[self.loc_id[i]]
            dr = host_reset_draw(self.rng, tb, self.day_lo[i], self.day_hi[i], self.steps)
            c0 = dr["c0"]
            for k in ("day", "hour", "ci_min", "ci_max", "t_min", "t_max"):
                ov[k][i] = dr[k]
            ov["t_win"][i] = dr["T"][c0:c0 + lw]
            ov["wb_win"][i] = dr["WB"][c0:c0 + lw]
            if i in self.oracles:
                lo, hi = max(0, c0 - 16), c0 + self.steps + 18
                NC = (tb["C"][lo:hi] - dr["ci_min"]) / (dr["ci_max"] - dr["ci_min"])
                NT = (dr["T"][lo:hi] - dr["t_min"]) / (dr["t_max"] - dr["t_min"])
                oobs[int(i)] = self.oracles[i].begin(tb["W"][lo:hi], tb["C"][lo:hi], NC, dr["T"][lo:hi], dr["WB"][lo:hi],
                                                     NT, lo, dr["day"], dr["hour"], self.steps)
        obs, _ = self.eng.reset(mask=mask.astype(np.uint8), override=ov)
        return G.raw_obs(obs.cpu().numpy()), oobs

    def step(self, actions_np):
        import torch
        a = torch.from_numpy(np.ascontiguousarray(actions_np, dtype=np.int32)).to(self.eng.device)
        obs, share, rew, done, info = self.eng.step(a)
        return (G.raw_obs(obs.cpu().numpy()), share.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(),
                info.cpu().numpy())


INFO_CMP = [k for k in po.INFO_COLS[:37] if not k.startswith("reserved")]


def compare_step(rig, acts, worst, check_every_env=True):
    eo, es, er, ed, ei = rig.step(acts)
    np.testing.assert_array_equal(es, G.share_from_raw(eo))   # share_obs is a re-arrangement of the same step's obs
    for i, orc in rig.oracles.items():
        oo, orew, odone, oinfo = orc.step(acts[i])
        worst["obs"] = max(worst["obs"], float(G.rel_err(eo[i], oo).max()))
        worst["rew"] = max(worst["rew"], float(G.rel_err(er[i], orew).max()))
        assert int(ed[i]) == odone
        for k in INFO_CMP:
            j = po.INFO_IDX[k]  # same column order in product and oracle for the first 37 columns
            worst["info"] = max(worst["info"], float(G.rel_err(ei[i, j], oinfo[j])))
        assert ei[i, L.INFO_IDX["fault"]] == oinfo[po.INFO_IDX["fault"]] == 0
    return ed


def run_engine_vs_oracle(n_envs=64, n_steps=200, episode_steps=672, seed=0, locations=("ny",),
                         dc_files=("dc_config.json",), oracle_envs=None, hist_cap=10000):
    rig = ParityRig(n_envs, episode_steps, seed, locations, dc_files, oracle_envs=oracle_envs, hist_cap=hist_cap)
    arng = np.random.default_rng(seed + 1)
    worst = dict(obs=0.0, rew=0.0, info=0.0)
    eobs, oobs = rig.reset_all()
    for i, o in oobs.items():
        worst["obs"] = max(worst["obs"], float(G.rel_err(eobs[i], o).max()))
    done_steps = 0
    for t in range(n_steps):
        acts = arng.integers(0, 3, size=(n_envs, 3)).astype(np.int32)
        ed = compare_step(rig, acts, worst)
        done_steps += 1
        if ed.any():
            assert ed.all()
            eobs, oobs = rig.reset_all()
            for i, o in oobs.items():
                worst["obs"] = max(worst["obs"], float(G.rel_err(eobs[i], o).max()))
    worst["env_steps"] = done_steps * len(rig.oracles)
    rig.eng.close()
    return worst
