"""Parity in the PRODUCTION configuration at the sizes and kernels the bench line quotes rates for.

One recipe, shared by tests/test_gpu_timed_config.py and tests/test_gpu_production_sizes.py: `debug_flags` as given (0 =
what bench.py times: no verify kernel), auto-reset with the device's own Philox resets, every history ring at its 10 000-
entry steady state (write positions spread, duplicates included), i.i.d. uniform actions, deferred window re-centring by
the spare wavefronts under the full request load -- and a SAMPLE of envs stepped by the fp64 oracle on the device's own
episode windows (read back after every reset).
Reference: sustaindc_env.py:533-621 (step), utils/reward_creator.py:16-45 (normalize_energy)."""
import numpy as np

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
from oracle import pyoracle as po
from tests import gpu_helpers as G
from tests.parity_util import INFO_CMP

TOL = 1e-5   # north_star: 1e-5 relative fp32 (absolute where |ref| < 1)
CAP = 10000
MIXED_FILES = ("dc_config.json", "dc_config_r16.json", "dc_config_r25.json")
MIXED_LOCATIONS = ("ny", "az", "wa")


def env_block_of_workgroup(b, n_blocks):
    """csrc/sdc_step.hip first_pair_of_block: workgroup b (after the sweep workgroups) -> the env block it steps."""
    return (b % 8) * (n_blocks // 8) + b // 8 if n_blocks % 8 == 0 else b


def sample_envs(N, envs_per_wave, rng, n_random=56):
    """Envs to check: every row of the first / last wavefronts, both ends of the batch, and the workgroups either side of
    every occupancy round of the launch (256 CUs: workgroups 255|256, 511|512, 767|768 in dispatch order -- at 16 384 envs
    with four envs per wavefront, 768.. is the fourth wavefront per SIMD that runs alone), plus a random spread."""
    epb = envs_per_wave * 4                      # envs per workgroup (4 wavefronts)
    nb = -(-N // epb)
    s = set(range(0, min(N, 2 * envs_per_wave))) | set(range(max(0, N - 2 * envs_per_wave), N))
    for b in (0, 1, 255, 256, 319, 320, 511, 512, 767, 768, 1023, nb - 1):
        if 0 <= b < nb:
            e0 = env_block_of_workgroup(b, nb) * epb
            s |= {e for e in (e0, e0 + 1, e0 + envs_per_wave - 1, e0 + envs_per_wave, e0 + epb - 1) if e < N}
    s |= set(int(x) for x in rng.choice(N, n_random, replace=False))
    return sorted(s)


class ProductionRig:
    """N envs on one engine in the production configuration + oracles for a sample of them."""

    def __init__(self, N, debug_flags=0, mixed=False, episode_steps=120, seed=77, envs_per_wave=2, n_random=56,
                 reward_method=(0, 0, 0), policy=(0, 0, 0), trim_and_respond_limit=27.0):
        self.N, self.steps = N, episode_steps
        rng = self.rng = np.random.default_rng(seed)
        locs = MIXED_LOCATIONS if mixed else ("ny",)
        files = MIXED_FILES if mixed else ("dc_config.json",)
        self.tables = [traces.synthetic_tables(loc, 0) for loc in locs]
        combos = [(li, f) for li in range(len(locs)) for f in files]
        self.params = [dc_config.size_datacenter(f, 1, traces.max_ambient_for_sizing(traces.obtain_paths(locs[li])[0]))
                       for li, f in combos]
        eng = self.eng = SdcEngine(N, episode_steps=episode_steps, auto_reset=True, seed=seed, debug_flags=debug_flags,
                                   n_locations=len(locs), n_dc_configs=len(combos), reward_method=reward_method, policy=policy,
                                   trim_and_respond_limit=trim_and_respond_limit)
        for li, tb in enumerate(self.tables):
            eng.set_tables(li, tb["W"], tb["C"], tb["T"], tb["WB"])
        for ci, p in enumerate(self.params):
            eng.set_dc_params(ci, p)
        e = np.arange(N)
        # BASELINE configs[3]: the rack count follows env_id % 3; the location changes every three envs
        self.loc_id = ((e // len(files)) % len(locs)).astype(np.int32)
        self.cfg_id = (self.loc_id * len(files) + e % len(files)).astype(np.int32)
        init_day = traces.get_init_day(6)
        eng.assign(self.loc_id, self.cfg_id, init_day - 7, init_day + 7)
        # steady-state history: every ring full, write positions spread, duplicates included
        hist = np.full((N, eng.hist_stride), np.nan, np.float32)
        vals = (331 + 70 * rng.standard_normal((N, CAP), dtype=np.float32)).clip(150, 650)
        vals[:, ::97] = vals[:, 5:6]
        hist[:, :CAP] = vals
        pos = rng.integers(0, CAP, N).astype(np.int32)
        eng.set_state("hist", hist)
        eng.set_state("hist_len", np.full(N, CAP, np.int32))
        eng.set_state("hist_pos", pos)
        del hist
        self.sample = sample_envs(N, envs_per_wave, rng, n_random)
        self.orcs = {}
        for i in self.sample:
            p = dict(self.params[self.cfg_id[i]], reward_method=tuple(int(m) for m in reward_method))
            o = po.OracleEnv(G.oracle_params_from_dict(p))
            o.e.stpt = float(p["init_setpoint"])
            o.e.hist_len = CAP
            o.e.hist_pos = int(pos[i])
            np.ctypeslib.as_array(o.e.hist)[:] = vals[i].astype(np.float64)
            self.orcs[i] = o
        del vals
        self.worst = dict(obs=0.0, rew=0.0, info=0.0)
        self.paths = np.zeros(8, np.int64)
        self.resets = 0

    def begin_all(self, obs_dev):
        """Start the oracles' next episode on the windows the DEVICE drew (read back); compares the reset observations."""
        eng, steps = self.eng, self.steps
        raw = G.raw_obs(obs_dev.cpu().numpy())
        st = {k: eng.get_state(k) for k in ("cursor", "day", "hourq", "t_min", "t_den", "ci_min", "ci_den")}
        tw, wb = eng.get_state("t_win"), eng.get_state("wb_win")
        for i, o in self.orcs.items():
            tb = self.tables[self.loc_id[i]]
            c0 = int(st["cursor"][i])
            lo, hi = max(0, c0 - 16), c0 + steps + 18
            T = np.zeros(hi - lo)
            WBv = np.zeros(hi - lo)
            T[c0 - lo:] = tw[i]
            WBv[c0 - lo:] = wb[i]
            NC = (tb["C"][lo:hi] - st["ci_min"][i]) / st["ci_den"][i]
            NT = (T - st["t_min"][i]) / st["t_den"][i]
            oo = o.begin(tb["W"][lo:hi], tb["C"][lo:hi], NC, T, WBv, NT, lo, int(st["day"][i]), int(st["hourq"][i]) // 4, steps)
            self.worst["obs"] = max(self.worst["obs"], float(G.rel_err(raw[i], oo).max()))

    def check_step(self, a_np, eo, er, ed, ei, fo):
        """One step's outputs (host arrays; eo / fo raw [N,53]) of the sampled envs against the oracle under actions a_np."""
        w = self.worst
        assert (ei[:, L.INFO_IDX["fault"]] == 0).all()
        self.paths += np.bincount(ei[:, L.INFO_IDX["reserved"]].astype(int), minlength=8)[:8]
        for i, o in self.orcs.items():
            oo, orew, odone, oinfo = o.step(a_np[i])
            assert int(ed[i]) == odone
            # at an episode end the step's own observation is in final_obs; obs already holds the next episode's first
            w["obs"] = max(w["obs"], float(G.rel_err(fo[i] if odone else eo[i], oo).max()))
            w["rew"] = max(w["rew"], float(G.rel_err(er[i], orew).max()))
            for k in INFO_CMP:
                j = po.INFO_IDX[k]      # same column order in product and oracle for the first 37 columns
                w["info"] = max(w["info"], float(G.rel_err(ei[i, j], oinfo[j])))

    def single_steps(self, n_steps, seed=78):
        import torch
        eng, N = self.eng, self.N
        arng = torch.Generator(device="cpu").manual_seed(seed)
        for t in range(n_steps):
            a_host = torch.randint(0, 3, (N, 3), dtype=torch.int32, generator=arng)
            obs, share, rew, done, info = eng.step(a_host.cuda())
            ed = done.cpu().numpy()
            fo = G.raw_obs(eng.final_obs.cpu().numpy()) if ed.any() else None
            self.check_step(a_host.numpy(), G.raw_obs(obs.cpu().numpy()), rew.cpu().numpy(), ed, info.cpu().numpy(), fo)
            if ed.any():
                assert ed.all()
                self.resets += 1
                self.begin_all(obs)

    def check_rollout(self, acts, out):
        """The K steps of a multi-step launch (obs [K,N,3,26], share, rew, done, info, ...) under actions acts [K,N,3]."""
        obs, share, rew, done, info = [x.cpu().numpy() for x in out[:5]]
        a = acts.cpu().numpy()
        fo = None
        for k in range(obs.shape[0]):
            if done[k].any():
                assert k == obs.shape[0] - 1 and done[k].all()
                fo = G.raw_obs(self.eng.final_obs.cpu().numpy())
            self.check_step(a[k], G.raw_obs(obs[k]), rew[k], done[k], info[k], fo)

    def assert_ok(self):
        w = self.worst
        assert w["obs"] <= TOL and w["rew"] <= TOL and w["info"] <= TOL, w
        assert (self.eng.get_state("hist_len") == CAP).all()

    def assert_all_reward_state_paths_seen(self):
        """The three ways a step's reward state is served all occurred: without a ring read, by taking over a deferred
        re-centred window, and with the ring read inside the step (rebuild after the injection / inline re-centring)."""
        p = self.paths
        assert p[0] > 0 and p[2] > 0 and p[1] + p[3] > 0, p
        assert p[0] + p[2] > 50 * (p[1] + p[3] - self.N), p   # (one rebuild per env right after the injection)
