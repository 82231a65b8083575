"""Episodes PREPARED AHEAD (sdc_capi.hip prep_arm / auto_reset_boundary): right after a boundary the reset + feature-row
kernels of the NEXT episode run on the library's second stream against shadow records and the alternate buffers while the
current episode is stepped; the boundary itself is a commit kernel.  Same kernels, same inputs: an engine with it
(the default) and one without (debug_flags bit 11) must agree bit for bit -- outputs AND state -- over many short episodes,
single steps, rollouts, the closed loop, and across everything that invalidates a prepared episode (masked / full reset,
sdc_set_state, sdc_set_seed).  Reference semantics: harl/envs/env_wrappers.py:176-190 (auto-reset inside the step call)."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine

pytestmark = pytest.mark.gpu
NO_PREP = 2048
STATE = ("record", "header", "qwin", "hist", "qtab", "t_win", "wb_win")


def _pair(N, steps, seed=3, flags=0, **kw):
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    out = []
    for f in (flags, flags | NO_PREP):
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=seed, debug_flags=f, **kw)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 170, 190)
        out.append(e)
    return out


def _same_state(a, b, where):
    for name in STATE:
        x, y = a.get_state(name), b.get_state(name)
        if name == "header":      # (a re-centring request's slot index is the order of an atomic)
            x[:, 34:38] &= ~np.uint32(0xFF)
            y[:, 34:38] &= ~np.uint32(0xFF)
        np.testing.assert_array_equal(x, y, err_msg=f"{name} {where}")


def _step_both(a, b, acts, t):
    for u, v, nm in zip(a.step(acts), b.step(acts), ("obs", "share_obs", "rew", "done", "info")):
        import torch
        assert torch.equal(u, v), (t, nm)
    import torch
    assert torch.equal(a.final_obs, b.final_obs), t


@pytest.mark.parametrize("N,flags", [(512, 0), (2600, 0), (512, 1)])     # (2600 envs: three prepare chunks; flags 1: verify mode)
def test_prepared_episodes_equal_synchronous_boundaries(N, flags):
    import torch
    steps = 40
    a, b = _pair(N, steps, flags=flags)
    oa, _ = a.reset()
    ob, _ = b.reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cpu").manual_seed(5)
    acts = torch.randint(0, 3, (6 * steps + 7, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(acts.shape[0]):                 # six boundaries
        _step_both(a, b, acts[t], t)
    sa, sb = a.boundary_stats(), b.boundary_stats()
    assert sa["prepared"] == 6 and sa["synchronous"] == 0 and sa["pending"], sa
    assert sb["prepared"] == 0 and sb["synchronous"] == 6 and not sb["pending"], sb
    _same_state(a, b, "after six boundaries")
    assert (a.get_state("episode") == 7).all() and (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    if flags == 0:
        # rollouts that end an episode, and the closed loop
        from tests.test_gpu_actor import _torch_actor
        for e in (a, b):
            for k in range(3):
                e.set_actor(k, _torch_actor(30 + k).state_dict())
        k = a.steps_to_episode_end()
        ra, rb = a.rollout(acts[:k].contiguous()), b.rollout(acts[:k].contiguous())
        for u, v in zip(ra, rb):
            assert torch.equal(u, v)
        assert a.steps_to_episode_end() == steps
        ra, rb = a.rollout_actor(steps, sample=True), b.rollout_actor(steps, sample=True)
        for u, v in zip(ra[:6], rb[:6]):
            assert torch.equal(u, v)
        assert a.boundary_stats()["prepared"] == 8
        _same_state(a, b, "after the multi-step launches")
    a.close()
    b.close()


def test_host_writes_invalidate_a_prepared_episode():
    import torch
    N, steps = 256, 32
    a, b = _pair(N, steps, seed=8)
    a.reset()
    b.reset()
    g = torch.Generator(device="cpu").manual_seed(6)
    acts = torch.randint(0, 3, (12 * steps, N, 3), dtype=torch.int32, generator=g).cuda()
    t = 0

    def run(n):
        nonlocal t
        for _ in range(n):
            _step_both(a, b, acts[t], t)
            t += 1

    run(steps + 5)                                              # boundary 1: prepared
    assert a.boundary_stats() == {"prepared": 1, "synchronous": 0, "pending": True}
    mask = (np.arange(N) % 5 == 0).astype(np.uint8)
    a.reset(mask=mask)                                          # a masked reset: the batch leaves lock-step, nothing is prepared
    b.reset(mask=mask)
    assert not a.boundary_stats()["pending"]
    run(steps)                                                  # both groups end an episode on the way (subset boundaries)
    assert a.boundary_stats()["prepared"] == 1
    a.reset()                                                   # back in lock-step: armed again
    b.reset()
    assert a.boundary_stats()["pending"]
    run(steps)
    assert a.boundary_stats()["prepared"] == 2
    a.set_seed(1234)                                            # the prepared episode was drawn under the old seed
    b.set_seed(1234)
    assert not a.boundary_stats()["pending"]
    run(steps)                                                  # -> synchronous, under the NEW seed (b does the same)
    st = a.boundary_stats()
    assert st["prepared"] == 2 and st["pending"]
    run(steps)
    assert a.boundary_stats()["prepared"] == 3
    sp = a.get_state("stpt")                                    # any host write to the state
    a.set_state("stpt", sp)
    b.set_state("stpt", sp)
    assert not a.boundary_stats()["pending"]
    run(2 * steps)
    assert a.boundary_stats()["prepared"] == 4
    # a checkpoint taken mid-episode and restored later brings the windows of ITS episode back (they are state)
    run(7)
    ck, ckb, t_ck = a.state_dict(), b.state_dict(), t
    ref = [x.clone() for x in a.step(acts[t])]
    b.step(acts[t])
    t += 1
    run(steps)                                                  # (crosses a boundary: the buffers have swapped roles since)
    a.load_state_dict(ck)
    b.load_state_dict(ckb)
    for u, v in zip(a.step(acts[t_ck]), ref):
        assert torch.equal(u, v)
    b.step(acts[t_ck])
    run(steps)
    a.close()
    b.close()


def test_prepare_ahead_with_the_engine_on_its_own_stream():
    """The caller's stream changes between calls (use_stream): the two events order the prepare against whichever stream
    a boundary runs on."""
    import torch
    N, steps = 512, 24
    a, b = _pair(N, steps, seed=4)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a.reset()
    b.reset()
    g = torch.Generator(device="cpu").manual_seed(9)
    acts = torch.randint(0, 3, (10 * steps, N, 3), dtype=torch.int32, generator=g).cuda()
    torch.cuda.synchronize()
    for t in range(acts.shape[0]):
        st = s1 if (t // 13) % 2 == 0 else s2
        a.use_stream(st)
        st.wait_stream(torch.cuda.current_stream())
        xa = a.step(acts[t])
        torch.cuda.current_stream().wait_stream(st)
        for u, v, nm in zip(xa, b.step(acts[t]), ("obs", "share_obs", "rew", "done", "info")):
            assert torch.equal(u, v), (t, nm)
        if t % 13 == 12:
            st.synchronize()
    assert a.boundary_stats()["prepared"] == 10
    _same_state(a, b, "two streams")
    a.close()
    b.close()
