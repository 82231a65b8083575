"""Soak test of the incremental reward state (four rank windows, running sums; csrc/sdc_trackers.hpp,
sdc_ringpath.hpp) in verify mode: thousands of steps over many envs, episode boundaries with device-side auto-reset,
a wrapping history ring, policy switches that change the shape of the energy distribution.  After every step the
verify kernel recomputes each env's order statistics by exact bisection and its clipped moments by a direct fp64 pass
over the ring and compares them with the incremental state and with the reported z-score (debug_flags bit 0)."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from tests import parity_util as P

pytestmark = pytest.mark.gpu


# (flags 1: verify mode; + 1024: the common-case kernel with four envs per wavefront -- the mapping of large batches -- whatever
# the batch size)
# (mixed: BASELINE configs[3] -- 16 / 20 / 25 racks x three locations -- on the common-case kernels, every env with its own copy
# of its config's scalars: round 4)
@pytest.mark.parametrize("hist_cap,steps_total,flags,mixed", [(10000, 2600, 1, False), (1500, 5200, 1, False),
                                                              (1500, 5200, 1 | 1024, False), (1500, 2600, 1, True)])
def test_reward_state_soak(hist_cap, steps_total, flags, mixed):
    import torch
    N, ep = 512, 288
    kw = dict(locations=("ny", "az", "wa"), dc_files=("dc_config.json", "dc_config_r16.json", "dc_config_r25.json")) if mixed else {}
    rig = P.ParityRig(N, episode_steps=ep, seed=77, hist_cap=hist_cap, with_oracle=False, debug_flags=flags, **kw)
    eng = rig.eng
    rng = np.random.default_rng(77)
    # start from a nearly full ring so that it wraps within the test; per-env spread, skew and level differ
    n0 = hist_cap - 40
    hist = np.full((N, eng.hist_stride), np.nan, np.float32)
    scale = 5.0 + 60.0 * rng.random((N, 1))
    base = rng.standard_normal((N, n0)) * scale
    skew = rng.random((N, 1)) < 0.5
    base = np.where(skew, np.abs(base) ** 1.3, base) + 40.0 * rng.standard_normal((N, 1))
    hist[:, :n0] = base.astype(np.float32)
    eng.set_state("hist", hist)
    eng.set_state("hist_len", np.full(N, n0, np.int32))
    eng.set_state("hist_pos", np.zeros(N, np.int32))
    rig.reset_all()
    eng_auto = eng   # ParityRig builds the engine with auto_reset off: reset explicitly at episode ends
    paths = np.zeros(4, np.int64)
    t_in_ep = 0
    for t in range(steps_total):
        phase = (t // 650) % 4
        a = torch.randint(0, 3, (N, 3), dtype=torch.int32, device=eng.device)
        if phase == 1:      # constant policy: narrow energy distribution, heavy relative tails
            a[:, 0] = 1
            a[:, 1] = 1
            a[:, 2] = 2
        elif phase == 2:    # half the envs constant
            a[::2, 0] = 1
            a[::2, 2] = 2
        elif phase == 3:    # always process the queue, always discharge
            a[:, 0] = 2
            a[:, 2] = 1
        obs, share, rew, done, info = eng_auto.step(a)
        t_in_ep += 1
        if t % 25 == 0 or t_in_ep == ep:
            inf = info.cpu().numpy()
            bad = np.nonzero(inf[:, L.INFO_IDX["fault"]])[0]
            assert bad.size == 0, (t, bad[:8], inf[bad[:8], L.INFO_IDX["fault"]])
            assert np.isfinite(rew.cpu().numpy()).all()
            paths += np.bincount(inf[:, 39].astype(int), minlength=4)[:4]
        if t_in_ep == ep:
            rig.reset_all()
            t_in_ep = 0
    # a mismatch at any step of any env leaves the sticky bit set
    assert (eng.get_state("order_stat_sticky") == 0).all()
    assert (eng.get_state("hist_len") == hist_cap).all()
    print("soak hist_cap", hist_cap, "paths sampled (no ring read, a window re-centred inline, a deferred re-centred "
          "window taken over, rebuilt):", paths)
    assert paths[0] > 0 and paths[1] + paths[2] > 0 and paths[3] > 0   # every way of serving a step was exercised
    assert paths[2] > 0                                     # the spare-wavefront re-centrings arrive and verify
    assert paths[3] < paths[1] + paths[2]                   # ... and rebuilds stay the exception
    eng.close()


@pytest.mark.parametrize("hist_cap,flags", [(33, 1), (64, 1), (65, 1), (100, 1), (200, 1), (64, 1 | 1024), (100, 1 | 1024)])
def test_rank_windows_small_histories(hist_cap, flags):
    """Histories about as long as a 64-key window, from empty: the windows list the whole history while it is shorter
    than a window, stay anchored at its ends when it is not much longer, and every eviction hits a window.  Random and
    constant policies (the latter: runs of equal keys).  Verify mode checks every window key against its rank."""
    import torch
    N, ep = 128, 96
    rig = P.ParityRig(N, episode_steps=ep, seed=91 + hist_cap, hist_cap=hist_cap, with_oracle=False, debug_flags=flags)
    eng = rig.eng
    rig.reset_all()
    paths = np.zeros(4, np.int64)
    steps = 3 * hist_cap + 2 * ep
    for t in range(steps):
        a = torch.randint(0, 3, (N, 3), dtype=torch.int32, device=eng.device)
        if (t // 48) % 3 == 1:
            a[:, 0] = 1
            a[:, 1] = 1
            a[:, 2] = 2
        obs, share, rew, done, info = eng.step(a)
        inf = info.cpu().numpy()
        bad = np.nonzero(inf[:, L.INFO_IDX["fault"]])[0]
        assert bad.size == 0, (t, bad[:8], inf[bad[:8], L.INFO_IDX["fault"]])
        assert np.isfinite(rew.cpu().numpy()).all()
        if t >= 3 * hist_cap:
            paths += np.bincount(inf[:, 39].astype(int), minlength=4)[:4]
        if (t + 1) % ep == 0:
            rig.reset_all()
    assert (eng.get_state("order_stat_sticky") == 0).all()
    assert (eng.get_state("hist_len") == hist_cap).all()
    print("hist_cap", hist_cap, "paths once the ring is full (windows only, re-centred, -, rebuilt):", paths)
    assert paths[3] * 20 < paths[0]      # a full ring is served by the windows, not by rebuilds
    eng.close()


def test_deferred_recentring_across_the_step_counter_wrap(monkeypatch):
    """The host's step counter (stamps of the re-centring requests) wraps at 3 * 2^22; start just below it and run across
    the wrap in verify mode: requests keep being served and taken over on both sides, nothing mismatches."""
    import torch
    monkeypatch.setenv("SDC_TEST_STEP_NO", str((3 << 22) - 160))     # (read only with debug_flags bit 6; each set_state below moves the counter on by 3)
    N, ep, cap = 512, 96, 1500
    rig = P.ParityRig(N, episode_steps=ep, seed=91, hist_cap=cap, with_oracle=False, debug_flags=1 | 64)
    eng = rig.eng
    rng = np.random.default_rng(91)
    hist = np.full((N, eng.hist_stride), np.nan, np.float32)
    hist[:, :cap] = (rng.standard_normal((N, cap)) * (10.0 + 40.0 * rng.random((N, 1)))).astype(np.float32)
    eng.set_state("hist", hist)
    eng.set_state("hist_len", np.full(N, cap, np.int32))
    eng.set_state("hist_pos", np.zeros(N, np.int32))
    rig.reset_all()
    before = after = 0
    for t in range(384):
        a = torch.randint(0, 3, (N, 3), dtype=torch.int32, device=eng.device)
        obs, share, rew, done, info = eng.step(a)
        inf = info.cpu().numpy()
        assert (inf[:, L.INFO_IDX["fault"]] == 0).all(), t
        taken = int((inf[:, 39] == 2).sum())
        if 20 <= t < 148:
            before += taken
        if t >= 170:
            after += taken
        if (t + 1) % ep == 0:
            rig.reset_all()
    assert (eng.get_state("order_stat_sticky") == 0).all()
    print("deferred take-overs before / after the wrap:", before, after)
    assert before > 0 and after > 0
    eng.close()
