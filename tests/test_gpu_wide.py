"""The lane-per-env step kernel (csrc/sdc_wide.hip, the largest batches; forced here by debug_flags bit 11) against the two-envs-per-
wavefront kernel (bit 9): same arithmetic in the same order, so every output and the state are the same BITS."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine

pytestmark = pytest.mark.gpu

WIDE, PAIR = 2048, 512


def _engines(N, steps, cfg="dc_config.json", seed=3, days=(200, 210), flags=(WIDE, PAIR)):
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter(cfg, 1, 30.0)
    engs = []
    for fl in flags:
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=seed, debug_flags=fl)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, *days)
        e.reset()
        engs.append(e)
    return engs


def _same_step(a, b, acts, t, what, skip_reserved=True):
    import torch
    rsv = L.INFO_IDX["reserved"]
    for u, v, nm in zip(a.step(acts), b.step(acts), ("obs", "share_obs", "rew", "done", "info")):
        if nm == "info" and skip_reserved:
            u, v = u.clone(), v.clone()
            u[:, rsv] = 0
            v[:, rsv] = 0
        if not torch.equal(u, v):
            bad = (u != v).nonzero()
            raise AssertionError((what, t, nm, bad[:6].tolist(), u[tuple(bad[0])].item(), v[tuple(bad[0])].item()))


@pytest.mark.parametrize("cfg", ["dc_config.json", "dc_config_r16.json", "dc_config_r25.json"])
def test_lane_per_env_kernel_equals_the_pair_kernel(cfg):
    """256 envs (four wavefronts), 96-step episodes, 230 steps = two auto-resets; 20 / 16 / 25 racks (the rack sums' tree has a
    different shape for each)."""
    import torch
    N, steps = 256, 96
    a, b = _engines(N, steps, cfg)
    g = torch.Generator(device="cpu").manual_seed(9)
    acts = torch.randint(0, 3, (230, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(230):
        _same_step(a, b, acts[t], t, cfg)
    # (25 racks: 11 distinct racks, more than the common-case form's class tables hold -- the kernel's general form takes that config)
    assert a.last_step_kernel() == ("sdc_dynamics_wide_kernel" if cfg != "dc_config_r25.json" else "sdc_dynamics_wide_gen_kernel")
    assert b.last_step_kernel() == "sdc_dynamics_fast_kernel"
    for name in ("record", "hist", "qtab"):
        np.testing.assert_array_equal(a.get_state(name), b.get_state(name), err_msg=name)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    a.close()
    b.close()


def test_lane_per_env_kernel_full_rings_whole_state():
    """1024 envs with full history rings (the timed configuration's steady state), 300 steps over three auto-resets: every output and
    the WHOLE state -- records, headers (running sums, window ranks, cached keys, request stamps), rank windows, rings, queue
    tables -- bit for bit against the pair kernel; every way a step's reward state is served occurs (a key inside a window, a
    bound crossing a key, requests filed, re-centred windows taken over)."""
    import torch
    N, steps, cap = 1024, 96, 10000
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    rng = np.random.default_rng(3)
    hist = np.full((N, 10240), np.nan, np.float32)
    hist[:, :cap] = (331 + 70 * rng.standard_normal((N, cap))).clip(150, 650).astype(np.float32)
    pos = rng.integers(0, cap, N).astype(np.int32)
    engs = []
    for flags in (WIDE, PAIR):
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=12, debug_flags=flags)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 174, 188)
        e.set_state("hist", hist)
        e.set_state("hist_len", np.full(N, cap, np.int32))
        e.set_state("hist_pos", pos)
        e.reset()
        engs.append(e)
    a, b = engs
    g = torch.Generator(device="cpu").manual_seed(5)
    acts = torch.randint(0, 3, (300, N, 3), dtype=torch.int32, generator=g).cuda()
    rsv = L.INFO_IDX["reserved"]
    paths = {"lane per env": {}, "two per wavefront": {}}
    took_over = np.zeros(N, dtype=bool)
    for t in range(300):
        _same_step(a, b, acts[t], t, "full rings")
        took_over |= (a.info[:, rsv] == 2).cpu().numpy() | (b.info[:, rsv] == 2).cpu().numpy()
        if t == 30:
            # (a re-centred window is the sweep's work: this kernel's one-wavefront sweeps (qt_refill) and the pair kernel's
            # four-wavefront ones (qt_refill_coop) centre on the same rank but may list a different number of keys beyond it -- both
            # valid.  Envs that took a re-centred window over are compared on what the windows ANSWER, the outputs; the whole
            # state bit for bit while none has.)
            assert not took_over.any()
            for name in ("record", "header", "qwin", "hist", "qtab"):
                sa, sb = a.get_state(name), b.get_state(name)
                if name == "header":      # (a request's slot index -- the low 11 bits of the four H_PEND words -- is the order of an atomic)
                    sa[:, 34:38] &= ~np.uint32(0x7FF)
                    sb[:, 34:38] &= ~np.uint32(0x7FF)
                bad = np.argwhere((sa != sb) & ~((sa != sa) & (sb != sb)))      # (empty ring slots read back as NaN)
                assert len(bad) == 0, (name, len(bad), bad[:8].tolist())
        for e, nm in ((a, "lane per env"), (b, "two per wavefront")):
            v, c = e.info[:, rsv].unique(return_counts=True)
            for x, k in zip(v.tolist(), c.tolist()):
                paths[nm][int(x)] = paths[nm].get(int(x), 0) + int(k)
    # info[reserved]: 0 incremental state only, 2 a re-centred window taken over, 1 a window re-centred inline, 3 rebuilt from the ring
    print("reward-state paths (env-steps by info[reserved]):", paths)
    assert a.last_step_kernel() == "sdc_dynamics_wide_kernel" and b.last_step_kernel() == "sdc_dynamics_fast_kernel"
    assert paths["lane per env"].get(2, 0) > 0          # re-centred windows arrived (requests were filed two steps earlier)
    # the whole-wavefront fallback with a ring read stays as rare as in the pair kernel (each is a ~5 us straggler of its launch)
    assert paths["lane per env"].get(1, 0) + paths["lane per env"].get(3, 0) <= paths["two per wavefront"].get(1, 0) + paths["two per wavefront"].get(3, 0) + 8, paths
    for name in ("record", "hist", "qtab"):
        np.testing.assert_array_equal(a.get_state(name), b.get_state(name), err_msg=name)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    a.close()
    b.close()


def test_the_host_picks_the_kernel_by_batch_size():
    """Single steps of a lock-step, single-config batch: two envs per wavefront up to 5 632 envs, four above, one lane per env from
    7 680 (sdc_capi.hip fast_case / quad_case / wide_case); debug_flags bit 12 keeps the lane-per-env kernel off."""
    import torch
    for N, flags, want in ((4096, 0, "sdc_dynamics_fast_kernel"), (7616, 0, "sdc_dynamics_quad_kernel"),
                           (7680, 0, "sdc_dynamics_wide_kernel"), (7680, 4096, "sdc_dynamics_quad_kernel"),
                           (9220, 0, "sdc_dynamics_quad_kernel")):
        (e,) = _engines(N, 96, flags=(flags,))
        e.step(torch.ones((N, 3), dtype=torch.int32, device="cuda"))
        assert e.last_step_kernel() == want, (N, flags, e.last_step_kernel())
        e.close()


def test_rollout_of_a_large_batch_is_single_step_launches_of_the_lane_per_env_kernel():
    """`sdc_rollout` over a batch the lane-per-env kernel serves (12 288 envs and up: sdc_capi.hip SDC_WIDE_ROLLOUT_MIN_ENVS): K launches of it inside the call, the same bits
    as K calls of step() on a twin engine and as the multi-step kernel of four envs per wavefront (debug_flags bit 12), across an
    episode end (auto-reset inside the last step)."""
    import torch
    N, steps, K = 12288, 48, 24
    a, b, c = _engines(N, steps, flags=(0, 0, 4096))
    g = torch.Generator(device="cpu").manual_seed(21)
    acts = torch.randint(0, 3, (2 * K, N, 3), dtype=torch.int32, generator=g).cuda()
    rsv = L.INFO_IDX["reserved"]
    for half in range(2):                       # the second rollout ends the episode
        seq = acts[half * K:(half + 1) * K].contiguous()
        ro = a.rollout(seq)
        rc = c.rollout(seq)
        assert a.last_step_kernel() == "sdc_dynamics_wide_kernel"
        for k in range(K):
            st = b.step(seq[k])
            for u, v, w, nm in zip(ro, st, rc, ("obs", "share_obs", "rew", "done", "info")):
                u, w = u[k], w[k]
                if nm == "info":
                    u, v, w = u.clone(), v.clone(), w.clone()
                    u[:, rsv] = 0; v[:, rsv] = 0; w[:, rsv] = 0
                assert torch.equal(u, v.reshape(u.shape)), (half, k, nm, "rollout vs step")
                assert torch.equal(u, w), (half, k, nm, "lane per env vs four per wavefront")
    assert bool(ro[3][-1].all())                # the episode ended in the last step of the second rollout
    for e in (a, b, c):
        assert (e.info[:, L.INFO_IDX["fault"]] == 0).all()
        e.close()


def test_the_queue_tables_time_major_mirror_is_kept_by_every_kernel():
    """The lane-per-env kernel reads its five queue-history probes from a time-major mirror of the queue table's `cum` column
    (SdcDev::qcum_t), which every kernel that appends to the table keeps.  (1) Engine A (lane per env) with a profiled step every
    third launch -- those run the general kernel (sdc_dynamics_kernel), which appends through pair_dynamics -- against a
    two-envs-per-wavefront engine: 150 steps of a 200-step episode (all probe lags live) bit-identical.  (2) A's state_dict() into a
    FRESH lane-per-env engine (a host write to the table rebuilds the mirror; the restored engine steps the general kernel until
    its next reset has recomputed the feature rows, then the lane-per-env kernel) and into a pair engine: all continue
    bit-identically over the episode end."""
    import torch
    N, steps = 256, 200
    a, b, c = _engines(N, steps, flags=(WIDE, WIDE, PAIR))
    g = torch.Generator(device="cpu").manual_seed(33)
    acts = torch.randint(0, 3, (260, N, 3), dtype=torch.int32, generator=g).cuda()
    rsv = L.INFO_IDX["reserved"]

    def same(outs, t, what):
        for nm, *xs in zip(("obs", "share_obs", "rew", "done", "info"), *outs):
            xs = [x.clone() for x in xs]
            if nm == "info":
                for x in xs:
                    x[:, rsv] = 0
            for x in xs[1:]:
                assert torch.equal(xs[0], x), (t, nm, what)

    a.profile(3)
    seen = set()
    for t in range(150):
        same([a.step(acts[t]), c.step(acts[t])], t, "profiled lane-per-env engine vs pair engine")
        seen.add(a.last_step_kernel())
    assert seen == {"sdc_dynamics_wide_kernel", "sdc_dynamics_kernel"}, seen
    a.profile(0)
    sd = a.state_dict()
    assert int(sd["qtab"].view(np.uint32).reshape(N, -1, 2)[:, 50:150, 0].max()) > 0      # (the table has content to mirror)
    b.load_state_dict(sd)
    c.load_state_dict(sd)
    kb = set()
    for t in range(150, 260):
        same([a.step(acts[t]), b.step(acts[t]), c.step(acts[t])], t, "restored engines")
        kb.add(b.last_step_kernel())
    assert kb == {"sdc_dynamics_kernel", "sdc_dynamics_wide_kernel"}, kb
    assert c.last_step_kernel() == "sdc_dynamics_fast_kernel"
    for e in (a, b, c):
        assert (e.info[:, L.INFO_IDX["fault"]] == 0).all()
        e.close()


def test_lane_per_env_kernel_in_verify_mode_from_empty_rings():
    """7 680 envs (the smallest batch the host gives to the lane-per-env kernel by itself) from EMPTY rings for 1 300 steps with
    debug_flags bit 0: after every step sdc_reward_verify_kernel checks every key of all four rank windows against its rank in the
    ring, the quartiles against an exact bisection and z against a direct fp64 pass.  Covers the young histories (fallback path for
    all 64 envs of a wavefront), the first rebuilds, the flood of re-centring requests around steps 64-100 and an auto-reset."""
    import torch
    N = 7680
    (e,) = _engines(N, 672, flags=(1,))
    g = torch.Generator(device="cpu").manual_seed(12)
    pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(1300):
        o, s, r, d, info = e.step(pool[t & 63])
        if t % 100 == 99:
            assert (info[:, L.INFO_IDX["fault"]] == 0).all(), t
            assert torch.isfinite(r).all(), t
    assert e.last_step_kernel() == "sdc_dynamics_wide_kernel"
    assert (e.get_state("order_stat_sticky") == 0).all()
    assert (e.info[:, L.INFO_IDX["fault"]] == 0).all()
    e.close()
