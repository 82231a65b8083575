"""The lane-per-env kernel's GENERAL form (csrc/sdc_wide.hip, template GEN -> `sdc_dynamics_wide_gen_kernel`; forced here by
debug_flags bit 11): every lane its own data-centre config (BASELINE configs[3]: 16 / 20 / 25 racks x three locations in ONE batch),
the rule-based policies of utils/rbc_agents.py:3-47, utils/trim_and_respond.py:8-38 and utils/base_agents.py inside the step, the
dc / battery agents under any of utils/reward_creator.py:154-334 -- against the two-envs-per-wavefront kernels (bit 9), which run the
same arithmetic in the same order: every output and the state are the same BITS.  (The pair kernels against the oracle and the
reference's fixtures: tests/test_gpu_golden.py, test_gpu_policies.py, test_gpu_tou.py; this form against the oracle at production
sizes: tests/test_gpu_production_sizes.py.)"""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine

pytestmark = pytest.mark.gpu

WIDE, PAIR = 2048, 512
FILES = ("dc_config.json", "dc_config_r16.json", "dc_config_r25.json")
LOCS = ("ny", "az", "wa")
GEN_KERNEL = "sdc_dynamics_wide_gen_kernel"


def _mixed_engines(N, steps, flags=(WIDE, PAIR), seed=3, full_rings=False, **kw):
    tabs = [traces.synthetic_tables(loc, 0) for loc in LOCS]
    combos = [(li, f) for li in range(len(LOCS)) for f in FILES]
    params = [dc_config.size_datacenter(f, 1, traces.max_ambient_for_sizing(traces.obtain_paths(LOCS[li])[0])) for li, f in combos]
    e_idx = np.arange(N)
    loc_id = ((e_idx // len(FILES)) % len(LOCS)).astype(np.int32)
    cfg_id = (loc_id * len(FILES) + e_idx % len(FILES)).astype(np.int32)
    rng = np.random.default_rng(seed)
    hist = pos = None
    if full_rings:
        cap = 10000
        hist = np.full((N, 10240), np.nan, np.float32)
        hist[:, :cap] = (331 + 70 * rng.standard_normal((N, cap))).clip(150, 650).astype(np.float32)
        pos = rng.integers(0, cap, N).astype(np.int32)
    engs = []
    for fl in flags:
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=seed, debug_flags=fl, n_locations=len(LOCS),
                      n_dc_configs=len(combos), **kw)
        for li, tb in enumerate(tabs):
            e.set_tables(li, tb["W"], tb["C"], tb["T"], tb["WB"])
        for ci, p in enumerate(params):
            e.set_dc_params(ci, p)
        e.assign(loc_id, cfg_id, 174, 188)
        if full_rings:
            e.set_state("hist", hist)
            e.set_state("hist_len", np.full(N, 10000, np.int32))
            e.set_state("hist_pos", pos)
        e.reset()
        engs.append(e)
    return engs


def _one_config_engines(N, steps, cfg="dc_config.json", flags=(WIDE, PAIR), seed=3, **kw):
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter(cfg, 1, 30.0)
    engs = []
    for fl in flags:
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=seed, debug_flags=fl, **kw)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 200, 210)
        e.reset()
        engs.append(e)
    return engs


def _same(outs_a, outs_b, t, what):
    import torch
    rsv = L.INFO_IDX["reserved"]
    for u, v, nm in zip(outs_a, outs_b, ("obs", "share_obs", "rew", "done", "info")):
        if nm == "info":
            u, v = u.clone(), v.clone()
            u[..., rsv] = 0
            v[..., rsv] = 0
        if not torch.equal(u, v):
            bad = (u != v).nonzero()
            raise AssertionError((what, t, nm, bad[:6].tolist(), u[tuple(bad[0])].item(), v[tuple(bad[0])].item()))


def _same_state(a, b, names=("record", "hist", "qtab")):
    for name in names:
        sa, sb = a.get_state(name), b.get_state(name)
        bad = np.argwhere((sa != sb) & ~((sa != sa) & (sb != sb)))
        assert len(bad) == 0, (name, len(bad), bad[:8].tolist())


def test_several_configs_in_one_batch_equal_the_pair_kernel():
    """BASELINE configs[3]: 384 envs (six wavefronts), nine (location, rack count) combinations interleaved env by env -- every
    wavefront carries all of them -- 96-step episodes, 230 steps = two auto-resets."""
    import torch
    N, steps = 384, 96
    a, b = _mixed_engines(N, steps)
    g = torch.Generator(device="cpu").manual_seed(9)
    acts = torch.randint(0, 3, (230, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(230):
        _same(a.step(acts[t]), b.step(acts[t]), t, "mixed configs")
    assert a.last_step_kernel() == GEN_KERNEL and b.last_step_kernel() == "sdc_dynamics_fast_kernel"
    _same_state(a, b)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    a.close()
    b.close()


def test_several_configs_full_rings_every_reward_path():
    """... with full history rings (the steady state the rates are quoted in), 1024 envs, 260 steps: windows updated in place, bound
    crossings, requests filed and re-centred windows taken over -- outputs bit for bit, and the whole state while no re-centred
    window has arrived (tests/test_gpu_wide.py says why only until then)."""
    import torch
    N, steps = 1024, 96
    a, b = _mixed_engines(N, steps, full_rings=True, seed=12)
    g = torch.Generator(device="cpu").manual_seed(5)
    acts = torch.randint(0, 3, (260, N, 3), dtype=torch.int32, generator=g).cuda()
    rsv = L.INFO_IDX["reserved"]
    took = 0
    for t in range(260):
        _same(a.step(acts[t]), b.step(acts[t]), t, "mixed configs, full rings")
        took += int((a.info[:, rsv] == 2).sum())
        if t == 8:      # (before the first re-centred window arrives: requests filed at step t are taken over at t + 2)
            assert took == 0
            for name in ("record", "header", "qwin", "hist", "qtab"):
                sa, sb = a.get_state(name), b.get_state(name)
                if name == "header":      # (a request's slot index is the order of an atomic)
                    sa[:, 34:38] &= ~np.uint32(0x7FF)
                    sb[:, 34:38] &= ~np.uint32(0x7FF)
                bad = np.argwhere((sa != sb) & ~((sa != sa) & (sb != sb)))
                assert len(bad) == 0, (name, len(bad), bad[:8].tolist())
    assert took > 0 and a.last_step_kernel() == GEN_KERNEL
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    a.close()
    b.close()


def test_a_config_with_more_classes_than_the_common_form_holds():
    """The 25-rack config alone has 11 distinct racks (the common-case form's tables hold 8): the general form serves it."""
    import torch
    N, steps = 256, 96
    a, b = _one_config_engines(N, steps, "dc_config_r25.json")
    g = torch.Generator(device="cpu").manual_seed(19)
    acts = torch.randint(0, 3, (150, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(150):
        _same(a.step(acts[t]), b.step(acts[t]), t, "r25")
    assert a.last_step_kernel() == GEN_KERNEL and b.last_step_kernel() == "sdc_dynamics_fast_kernel"
    _same_state(a, b)
    a.close()
    b.close()


@pytest.mark.parametrize("policy", [(1, 3, 2), (0, 3, 0), (1, 1, 2), (0, 0, 2)])
def test_rule_based_policies_inside_the_step(policy):
    """sdc_config.policy (0 external, 1 do nothing, 2 RBCBatteryAgent, 3 trim and respond): single steps and `sdc_rollout` -- with and
    without an action array -- on the general form against the pair kernels; the actions the policies chose come back through
    `actions_out`, the trim-and-respond counter lives in the record."""
    import torch
    N, steps = 256, 96
    all_pol = all(p != 0 for p in policy)
    a, b = _one_config_engines(N, steps, policy=policy, trim_and_respond_limit=34.9)
    g = torch.Generator(device="cpu").manual_seed(29)
    acts = torch.randint(0, 3, (steps + 40, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(40):                                    # single steps (sdc_step)
        x = None if all_pol and t % 2 else acts[t]
        _same(a.step(x), b.step(x), t, ("steps", policy))
    assert a.last_step_kernel() == GEN_KERNEL and b.last_step_kernel() == "sdc_dynamics_kernel"
    k = a.steps_to_episode_end()
    assert k == steps - 40
    seq = None if all_pol else acts[40:40 + k].contiguous()
    ra = a.rollout_policy(k, seq)                          # K launches of the general form, the episode's end inside
    rb = b.rollout_policy(k, seq)
    assert a.last_step_kernel() == GEN_KERNEL and b.last_step_kernel() == "sdc_rollout_kernel"
    _same(ra[:5], rb[:5], 0, ("rollout", policy))
    assert torch.equal(ra[5], rb[5])                       # the applied actions
    if policy[1] == 3:
        assert len(ra[5][:, :, 1].unique()) >= 2           # (the controller did respond)
    if policy[2] == 2:
        assert set(ra[5][:, :, 2].unique().tolist()) == {0, 1}
    for t in range(20):                                    # ... and on into the next episode
        x = None if all_pol else acts[t]
        _same(a.step(x), b.step(x), t, ("after the reset", policy))
    _same_state(a, b)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    a.close()
    b.close()


@pytest.mark.parametrize("methods", [(0, 3, 6), (0, 4, 5), (0, 1, 2), (0, 6, 3)])
def test_other_reward_functions_for_the_dc_and_battery_agents(methods):
    """sdc_config.reward_method (1 footprint, 2 custom = 0, 3 tou_reward, 4 energy_efficiency_reward, 5 energy_PUE_reward,
    6 water_usage_efficiency_reward: utils/reward_creator.py:154-334) on the dc / battery slots, several configs in the batch, full
    rings; the ls agent keeps default_ls_reward (with another one the history is not appended to: the pair kernels' job)."""
    import torch
    N, steps = 384, 96
    a, b = _mixed_engines(N, steps, full_rings=True, seed=40 + methods[1], reward_method=methods)
    g = torch.Generator(device="cpu").manual_seed(31)
    acts = torch.randint(0, 3, (200, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(200):
        _same(a.step(acts[t]), b.step(acts[t]), t, methods)
    assert a.last_step_kernel() == GEN_KERNEL and b.last_step_kernel() == "sdc_dynamics_kernel"
    if 3 in methods or 6 in methods:
        assert float(a.rew[:, 1:].abs().max()) > 0
    a.close()
    b.close()


def test_another_ls_reward_function_stays_on_the_pair_kernels():
    """reward_method[0] != default: the energy history is not appended to -- not a mode of the per-lane reward path."""
    import torch
    N = 256
    (a,) = _one_config_engines(N, 96, flags=(WIDE,), reward_method=(3, 0, 0))
    a.step(torch.ones((N, 3), dtype=torch.int32, device="cuda"))
    assert a.last_step_kernel() == "sdc_dynamics_kernel"
    a.close()
