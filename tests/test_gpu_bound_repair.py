"""env_reward's repair of "a clip bound left its window" (csrc/sdc_pairstep.hpp: the bounds' tail sums redone from the ring in one
pass instead of a full rebuild; the window moved by the ahead-of-need refill) -- a path ~4e-8 of the env-steps take by themselves.
debug_flags bit 13 makes every 61st (env + launch) take it; bit 0 (verify mode) has sdc_reward_verify_kernel check, after every
step, every key of all four rank windows against its rank in the ring, the quartiles against an exact bisection and the reported
z-score against a direct fp64 pass over the env's history."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from tests import parity_util as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hist_cap", [10000, 1500])
def test_forced_bound_repairs_verify_and_match_the_unforced_run(hist_cap):
    import torch
    N, ep, steps_total = 256, 288, 1500
    rigs = [P.ParityRig(N, episode_steps=ep, seed=5, hist_cap=hist_cap, with_oracle=False, debug_flags=f) for f in (1 | 8192, 1)]
    rng = np.random.default_rng(5)
    n0 = hist_cap - 200          # the ring fills and wraps inside the test
    hist = np.full((N, rigs[0].eng.hist_stride), np.nan, np.float32)
    scale = 5.0 + 60.0 * rng.random((N, 1))
    base = rng.standard_normal((N, n0)) * scale
    base = np.where(rng.random((N, 1)) < 0.5, np.abs(base) ** 1.3, base) + 40.0 * rng.standard_normal((N, 1))
    hist[:, :n0] = base.astype(np.float32)
    for r in rigs:
        r.eng.set_state("hist", hist)
        r.eng.set_state("hist_len", np.full(N, n0, np.int32))
        r.eng.set_state("hist_pos", np.zeros(N, np.int32))
        r.reset_all()
    g = torch.Generator(device="cpu").manual_seed(5)
    repaired = 0
    worst = 0.0
    t_in_ep = 0
    for t in range(steps_total):
        a = torch.randint(0, 3, (N, 3), dtype=torch.int32, generator=g).to(rigs[0].eng.device)
        if (t // 300) % 2 == 1:      # a constant policy for a while: a narrow energy distribution, bounds that move a lot
            a[:, 0] = 1
            a[:, 2] = 2
        outs = [r.eng.step(a) for r in rigs]
        t_in_ep += 1
        (obs_f, _, rew_f, _, info_f), (obs_u, _, rew_u, _, info_u) = outs
        assert torch.equal(obs_f, obs_u), t
        if t % 10 == 0 or t_in_ep == ep:
            inf = info_f.cpu().numpy()
            assert not inf[:, L.INFO_IDX["fault"]].any(), (t, np.nonzero(inf[:, L.INFO_IDX["fault"]])[0][:8])
            assert not info_u.cpu().numpy()[:, L.INFO_IDX["fault"]].any(), t
            repaired += int((inf[:, L.INFO_IDX["reserved"]] == 5).sum())
            # fresh totals against running sums: the same z to rounding
            d = (rew_f - rew_u).abs().max().item()
            worst = max(worst, d)
            assert d <= 1e-9 * max(1.0, rew_u.abs().max().item()), (t, d)
        if t_in_ep == ep:
            for r in rigs:
                r.reset_all()
            t_in_ep = 0
    for r in rigs:
        assert (r.eng.get_state("order_stat_sticky") == 0).all()      # verify mode saw no mismatch at any step of any env
        r.eng.close()
    print("bound repairs sampled:", repaired, "worst reward difference against the unforced run:", worst)
    assert repaired > 100
