"""Parity in the configuration bench.py times, and the reference's real-trace fixtures as ONE heterogeneous batch.

Every other oracle / golden parity test builds its engine in verify mode (debug_flags bit 0: a checking kernel after
each step).  These two run the PRODUCTION configuration -- debug_flags = 0, and for the first one everything else the
timed region has too: 4096 envs (BASELINE.json configs[2]), auto-reset with the device's own Philox resets, every history
ring at its 10 000-entry steady state, i.i.d. uniform actions, deferred window re-centring by the spare wavefronts under
the full request load.
Reference: sustaindc_env.py:533-621 (step), utils/reward_creator.py:16-45 (normalize_energy)."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
from tests import gpu_helpers as G
from tests.conftest import golden_names
from tests.test_gpu_golden import EXACT_INFO, assert_obs_bit_identical

pytestmark = pytest.mark.gpu

TOL = 1e-5   # north_star: 1e-5 relative fp32 (absolute where |ref| < 1)


def test_timed_configuration_4096_envs_vs_oracle():
    """4096 envs, debug_flags = 0, auto_reset, all rings full (10 000), i.i.d. actions, 330 steps over 3 episodes
    (2 auto-resets): 72+ sampled envs against the oracle at 1e-5 on every step, and every reward-state path seen.
    (The recipe lives in tests/production_rig.py; tests/test_gpu_production_sizes.py runs it at the other sizes / kernels
    the bench line quotes.)"""
    from tests.production_rig import ProductionRig
    rig = ProductionRig(4096, debug_flags=0, episode_steps=120, seed=77, envs_per_wave=2)
    assert len(rig.sample) >= 64
    obs, _ = rig.eng.reset()
    rig.begin_all(obs)
    rig.single_steps(330)
    print("timed configuration:", rig.worst, "reward-state paths (0 none / 1 inline / 2 deferred / 3 rebuilt):",
          rig.paths[:4], "auto-resets:", rig.resets)
    assert rig.resets >= 2
    rig.assert_ok()
    rig.assert_all_reward_state_paths_seen()
    rig.eng.close()


def test_golden_fixtures_as_one_heterogeneous_batch():
    """Every single-env episode fixture captured from the reference (real trace windows: 8 locations, 16 / 20 / 25 racks,
    0.5 / 1 / 2 MW, eight policies) as a DIFFERENT env of ONE engine: per-env location tables, data-centre config,
    start cursor and weather windows in one launch, production configuration (debug_flags = 0).  Observations
    bit-identical (documented exception: test_gpu_golden), rewards 1e-5, info 2e-6, integer info exact.
    Engine-wide settings the fixtures must share: the default reward functions (two fixtures use alternates: excluded)
    and the episode length (672 steps: the 30-day fixture contributes its first 672 steps, the 16-episode one its first
    episode)."""
    import torch
    names = [n for n in golden_names() if "meta_reward_method" not in G.load_fixture(n).files]
    assert len(names) >= 13
    fx = [G.load_fixture(n) for n in names]
    N, steps = len(fx), 672
    eng = SdcEngine(N, episode_steps=steps, auto_reset=False, n_locations=N, n_dc_configs=N, debug_flags=0)
    z = np.zeros(G.TL)
    for i, d in enumerate(fx):
        W, Cc = G.tables_from_fixture(d, 1)
        eng.set_tables(i, W, Cc, z, z)
        eng.set_dc_params(i, G.params_from_fixture(d))
    ids = np.arange(N, dtype=np.int32)
    eng.assign(ids, ids, 0, 364)
    ov = None
    for i, d in enumerate(fx):
        lw_fix = min(eng.lw, len(d["ep0_T"]) - (int(d["ep0_cursor0"]) - int(d["ep0_win_lo"])))
        assert lw_fix == eng.lw
        o = G.override_from_fixture(d, 0, 1, eng.lw)
        if ov is None:
            ov = {k: [v] for k, v in o.items()}
        else:
            for k, v in o.items():
                ov[k].append(v)
    ov = {k: np.concatenate(v, axis=0) for k, v in ov.items()}
    obs, share = eng.reset(override=ov)
    raw = G.raw_obs(obs.cpu().numpy())
    for i, d in enumerate(fx):
        assert int(eng.get_state("cursor")[i]) == int(d["ep0_cursor0"])
        assert_obs_bit_identical(raw[i][None], d["ep0_reset_obs"][None], int(d["ep0_cursor0"]) - 1, (names[i], "reset"))
    acts = torch.from_numpy(np.stack([d["ep0_actions"][:steps] for d in fx], axis=1).astype(np.int32)).cuda()   # [steps, N, 3]
    O = np.zeros((steps, N, 53), np.float32)
    R = np.zeros((steps, N, 3), np.float32)
    D = np.zeros((steps, N), np.uint8)
    I = np.zeros((steps, N, L.INFO_DIM), np.float32)
    for t in range(steps):
        obs, share, rew, done, info = eng.step(acts[t].contiguous())
        O[t] = G.raw_obs(obs.cpu().numpy())
        R[t] = rew.cpu().numpy()
        D[t] = done.cpu().numpy()
        I[t] = info.cpu().numpy()
    worst = dict(rew=0.0, info=0.0)
    for i, d in enumerate(fx):
        keys = [str(k) for k in d["meta_info_keys"]]
        assert_obs_bit_identical(O[:, i], d["ep0_obs"][:steps], int(d["ep0_cursor0"]), names[i])
        er = G.rel_err(R[:, i], d["ep0_rew"][:steps])
        worst["rew"] = max(worst["rew"], float(er.max()))
        assert er.max() <= TOL, (names[i], "rew", er.max())
        gd = d["ep0_done"][:steps].copy()
        gd[-1] = 1                      # (this engine's episodes end at step 672; so do all fixtures' but the 30-day one)
        np.testing.assert_array_equal(D[:, i], gd, err_msg=names[i])
        for j, k in enumerate(keys):
            col, ref = I[:, i, L.INFO_IDX[k]], d["ep0_info"][:steps, j]
            if k in EXACT_INFO:
                np.testing.assert_array_equal(col, ref.astype(np.float32), err_msg=f"{names[i]} {k}")
            else:
                ei = G.rel_err(col, ref)
                worst["info"] = max(worst["info"], float(ei.max()))
                assert ei.max() <= 2e-6, (names[i], k, int(ei.argmax()), ei.max())
        assert (I[:, i, L.INFO_IDX["fault"]] == 0).all(), names[i]
    print("heterogeneous golden batch:", len(names), "fixtures", names, worst)
    eng.close()


def test_common_case_kernels_equal_the_general_kernels():
    """The step / rollout kernels specialised for the common case (lock-step batch with feature rows, one config, external
    actions, default rewards, no diagnostics: what bench.py times) -- four envs per wavefront (debug_flags bit 10: the default
    for large batches) and two (bit 9) -- against the general kernels (debug_flags bit 7 forces them): every
    output and the whole state bit for bit, over auto-resets, single steps and rollouts."""
    import torch
    N, steps, cap = 1024, 96, 10000
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    rng = np.random.default_rng(3)
    hist = np.full((N, 10240), np.nan, np.float32)
    hist[:, :cap] = (331 + 70 * rng.standard_normal((N, cap))).clip(150, 650).astype(np.float32)
    pos = rng.integers(0, cap, N).astype(np.int32)
    engs = []
    for flags in (1024, 512, 128):       # four envs per wavefront, two, the general kernels
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=12, debug_flags=flags)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 174, 188)
        e.set_state("hist", hist)
        e.set_state("hist_len", np.full(N, cap, np.int32))
        e.set_state("hist_pos", pos)
        e.reset()
        engs.append(e)
    a = engs[0]
    g = torch.Generator(device="cpu").manual_seed(5)
    acts = torch.randint(0, 3, (260, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(200):                       # two auto-resets
        xa = a.step(acts[t])
        for b, which in zip(engs[1:], ("two envs per wavefront", "general")):
            xb = b.step(acts[t])
            for u, v, nm in zip(xa, xb, ("obs", "share_obs", "rew", "done", "info")):
                assert torch.equal(u, v), (t, nm, which, (u != v).nonzero()[:4].tolist())
    k = min(48, a.steps_to_episode_end())
    ra = a.rollout(acts[200:200 + k])
    sa = {name: a.get_state(name) for name in ("record", "header", "qwin", "hist", "qtab")}
    # (a re-centring request's slot index -- the low byte of the four H_PEND words -- is the order of an atomic)
    sa["header"][:, 34:38] &= ~np.uint32(0xFF)
    for b, which in zip(engs[1:], ("two envs per wavefront", "general")):
        assert torch.equal(a.final_obs, b.final_obs), which
        rb = b.rollout(acts[200:200 + k])
        for u, v in zip(ra, rb):
            assert torch.equal(u, v), which
        for name in sa:
            sb = b.get_state(name)
            if name == "header":
                sb[:, 34:38] &= ~np.uint32(0xFF)
            np.testing.assert_array_equal(sa[name], sb, err_msg=name + " / " + which)
    for e in engs:
        e.close()


@pytest.mark.parametrize("N", [7168, 7176])      # (7176: a partly filled last workgroup, a grid that is not a multiple of 8; from 7 680 envs: one lane per env)
def test_large_batches_take_the_four_env_mapping(N):
    """Large batches (single steps above 5632 envs and below 7680, the multi-step kernels above 4096) run four envs per wavefront by default (sdc_capi.hip quad_case): same outputs
    and state, bit for bit, as the two-env mapping (debug_flags bit 9) at those sizes -- steps, a rollout, the closed loop."""
    import torch
    from tests.test_gpu_actor import _torch_actor
    steps = 96
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    engs = []
    for flags in (0, 512):
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=21, debug_flags=flags)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 100, 120)
        for a in range(3):
            e.set_actor(a, _torch_actor(40 + a, "tanh").state_dict())
        e.reset()
        engs.append(e)
    a, b = engs
    g = torch.Generator(device="cpu").manual_seed(8)
    acts = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(40):
        for u, v, nm in zip(a.step(acts[t]), b.step(acts[t]), ("obs", "share_obs", "rew", "done", "info")):
            assert torch.equal(u, v), (t, nm)
    for u, v in zip(a.rollout(acts[40:56]), b.rollout(acts[40:56])):
        assert torch.equal(u, v)
    ra, rb = a.rollout_actor(24, sample=True), b.rollout_actor(24, sample=True)
    for u, v in zip(ra[:6], rb[:6]):
        assert torch.equal(u, v)
    for name in ("record", "qwin", "hist", "qtab"):
        np.testing.assert_array_equal(a.get_state(name), b.get_state(name), err_msg=name)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    for e in engs:
        e.close()


@pytest.mark.parametrize("cfg", ["dc_config_r16.json", "dc_config_r25.json"])
def test_four_env_mapping_other_rack_counts(cfg):
    """16 racks fill exactly one pass of the four-envs-per-wavefront rack model (16 lanes per env), 25 need the second
    pass for 9: both against the two-env mapping (one pass of 32 lanes), bit for bit."""
    import torch
    N, steps = 256, 96
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter(cfg, 1, 30.0)
    engs = []
    for flags in (1024, 512):
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=3, debug_flags=flags)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 200, 210)
        e.reset()
        engs.append(e)
    a, b = engs
    g = torch.Generator(device="cpu").manual_seed(9)
    acts = torch.randint(0, 3, (120, N, 3), dtype=torch.int32, generator=g).cuda()
    # (histories start empty here: around step 64-100 every env's young windows are re-centred at once, more requests than
    # the 128 slots of a step -- which envs get a slot and which re-centre inline is the order of an atomic, so the
    # diagnostic `reserved` column and the windows' centring may differ between the mappings; what they compute may not)
    rsv = L.INFO_IDX["reserved"]
    for t in range(120):                 # one auto-reset inside
        for u, v, nm in zip(a.step(acts[t]), b.step(acts[t]), ("obs", "share_obs", "rew", "done", "info")):
            if nm == "info":
                u, v = u.clone(), v.clone()
                u[:, rsv] = 0
                v[:, rsv] = 0
                assert torch.equal(u, v), (cfg, t, nm, (u != v).nonzero()[:4].tolist())
            else:
                assert torch.equal(u, v), (cfg, t, nm)
    for name in ("record", "hist"):
        np.testing.assert_array_equal(a.get_state(name), b.get_state(name), err_msg=name)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    for e in engs:
        e.close()


def test_common_case_kernels_serve_several_configs():
    """BASELINE configs[3] (16 / 20 / 25 racks by env_id % 3, three locations) on the kernels specialised for the common case
    (round 4: every env carries its own copy of its config's scalars, SdcDev::prm_env, so that they arrive with the record)
    against the general kernels (debug_flags bit 7): every output and the whole state bit for bit over single steps, two
    auto-resets and a rollout, rings full -- and after the host re-assigns the configs."""
    import torch
    from tests.production_rig import MIXED_FILES, MIXED_LOCATIONS
    N, steps, cap = 1026, 96, 10000       # (1026: a partly filled last workgroup)
    tabs = [traces.synthetic_tables(loc, 0) for loc in MIXED_LOCATIONS]
    combos = [(li, f) for li in range(3) for f in MIXED_FILES]
    params = [dc_config.size_datacenter(f, 1, traces.max_ambient_for_sizing(traces.obtain_paths(MIXED_LOCATIONS[li])[0]))
              for li, f in combos]
    e_idx = np.arange(N)
    loc_id = ((e_idx // 3) % 3).astype(np.int32)
    cfg_id = (loc_id * 3 + e_idx % 3).astype(np.int32)
    rng = np.random.default_rng(4)
    hist = np.full((N, 10240), np.nan, np.float32)
    hist[:, :cap] = (331 + 70 * rng.standard_normal((N, cap))).clip(150, 650).astype(np.float32)
    pos = rng.integers(0, cap, N).astype(np.int32)
    engs = []
    for flags in (0, 128):
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=14, debug_flags=flags, n_locations=3, n_dc_configs=9)
        for li, tb in enumerate(tabs):
            e.set_tables(li, tb["W"], tb["C"], tb["T"], tb["WB"])
        for ci, p in enumerate(params):
            e.set_dc_params(ci, p)
        e.assign(loc_id, cfg_id, 174, 188)
        e.set_state("hist", hist)
        e.set_state("hist_len", np.full(N, cap, np.int32))
        e.set_state("hist_pos", pos)
        e.reset()
        engs.append(e)
    a, b = engs
    g = torch.Generator(device="cpu").manual_seed(6)
    acts = torch.randint(0, 3, (300, N, 3), dtype=torch.int32, generator=g).cuda()

    def same_steps(t0, t1):
        for t in range(t0, t1):
            for u, v, nm in zip(a.step(acts[t]), b.step(acts[t]), ("obs", "share_obs", "rew", "done", "info")):
                assert torch.equal(u, v), (t, nm, (u != v).nonzero()[:4].tolist())

    same_steps(0, 200)                                # two auto-resets
    k = min(40, a.steps_to_episode_end())
    for u, v in zip(a.rollout(acts[200:200 + k]), b.rollout(acts[200:200 + k])):
        assert torch.equal(u, v)
    # the host moves every env to another rack count: the envs' copies of the scalars follow (sdc_set_state "cfg_id")
    cfg2 = (loc_id * 3 + (e_idx + 1) % 3).astype(np.int32)
    for e in engs:
        e.set_state("cfg_id", cfg2)
        e.reset()
    same_steps(240, 290)
    for name in ("record", "header", "qwin", "hist", "qtab"):
        x, y = a.get_state(name), b.get_state(name)
        if name == "header":
            x[:, 34:38] &= ~np.uint32(0xFF)
            y[:, 34:38] &= ~np.uint32(0xFF)
        np.testing.assert_array_equal(x, y, err_msg=name)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    for e in engs:
        e.close()
