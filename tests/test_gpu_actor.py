"""The closed loop in one launch (sdc_rollout_actor): the three agents' actor networks evaluated INSIDE the rollout kernel.

  * the in-kernel network against a plain PyTorch fp32 restatement of the reference's StochasticPolicy
    (harl/models/policy_models/stochastic_policy.py:11-60: LayerNorm(26) -> Linear(26,64) -> tanh -> LayerNorm(64) ->
    Linear(64,64) -> tanh -> LayerNorm(64) -> Linear(64,3) -> Categorical), same weights, evaluated on the very
    observations the kernel saw: logits within 2e-4 (fp32, different summation order, hardware exp2 / rsq), actions equal
    wherever the top-two logit gap exceeds that;
  * the dynamics under those actions against the external-action rollout of a second engine: bit for bit;
  * sampling: the empirical action frequencies against softmax(logits)."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine

pytestmark = pytest.mark.gpu


def _torch_actor(seed, activation="tanh"):
    import torch
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    act = nn.Tanh if activation == "tanh" else nn.ReLU

    class Base(nn.Module):
        def __init__(self):
            super().__init__()
            self.feature_norm = nn.LayerNorm(26)
            self.mlp = nn.Module()
            self.mlp.fc = nn.Sequential(nn.Linear(26, 64), act(), nn.LayerNorm(64), nn.Linear(64, 64), act(), nn.LayerNorm(64))

    class Policy(nn.Module):      # parameter names as in the reference's StochasticPolicy
        def __init__(self):
            super().__init__()
            self.base = Base()
            self.act = nn.Module()
            self.act.action_out = nn.Module()
            self.act.action_out.linear = nn.Linear(64, 3)

        def forward(self, x):
            return self.act.action_out.linear(self.base.mlp.fc(self.base.feature_norm(x)))

    m = Policy()
    with torch.no_grad():        # trained-looking weights: every parameter random, LayerNorm affine included
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.35 if p.dim() == 2 else 0.2) + (1.0 if p.dim() == 1 and p.numel() in (26, 64) and False else 0.0))
        for ln in (m.base.feature_norm, m.base.mlp.fc[2], m.base.mlp.fc[5]):
            ln.weight.copy_(1.0 + 0.2 * torch.randn(ln.weight.shape, generator=g))
    return m


def _engine(N, steps, seed=5, debug_flags=0):
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
    e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=seed, debug_flags=debug_flags)
    e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
    e.set_dc_params(0, p)
    e.assign(0, 0, 174, 188)
    return e


# (40 envs: a partly filled workgroup; debug_flags 1024: four envs per wavefront -- the mapping of large batches -- whatever the size)
@pytest.mark.parametrize("activation,N,flags", [("tanh", 512, 0), ("relu", 512, 0), ("tanh", 40, 0), ("tanh", 512, 1024),
                                                ("relu", 40, 1024)])
def test_in_kernel_actor_matches_torch_and_external_rollout(activation, N, flags):
    import torch
    steps, K = 96, 40
    nets = [_torch_actor(100 + a, activation) for a in range(3)]
    a_eng, b_eng = _engine(N, steps, debug_flags=flags), _engine(N, steps)
    for a in range(3):
        sd = dict(nets[a].state_dict())
        sd["activation"] = activation
        a_eng.set_actor(a, sd)
    obs0, _ = a_eng.reset()
    obs0 = obs0.clone()
    b_eng.reset()
    worst, checked = 0.0, 0
    prev_obs = obs0
    for rnd in range(3):          # 3 launches: 40 + 40 + 16 steps, the last one ends the episode (auto-reset inside)
        k = min(K, a_eng.steps_to_episode_end())
        obs, share, rew, done, info, acts, logits = a_eng.rollout_actor(k, sample=False, want_logits=True)
        # the networks, step by step, on the observations the kernel chose from
        inp = torch.cat([prev_obs[None], obs[:-1]], 0)                       # [k, N, 3, 26]
        for a in range(3):
            with torch.no_grad():
                ref = nets[a](inp[:, :, a, :].cpu()).numpy()                 # [k, N, 3]
            got = logits[:, :, a, :].cpu().numpy()
            err = np.abs(got - ref).max()
            worst = max(worst, float(err))
            assert err <= 2e-4, (rnd, a, err)
            top2 = np.sort(ref, axis=-1)
            clear = (top2[..., 2] - top2[..., 1]) > 1e-3
            np.testing.assert_array_equal(acts[:, :, a].cpu().numpy()[clear], ref.argmax(-1)[clear])
            checked += int(clear.sum())
        # the same actions fed from outside give the same trajectory, bit for bit
        ob, sb, rb, db, ib = b_eng.rollout(acts)
        assert torch.equal(obs, ob) and torch.equal(rew, rb) and torch.equal(done, db) and torch.equal(share, sb)
        ia, ibb = info.clone(), ib.clone()
        ia[..., L.INFO_IDX["reserved"]] = 0
        ibb[..., L.INFO_IDX["reserved"]] = 0
        assert torch.equal(ia, ibb)
        prev_obs = obs[-1].clone()
    assert a_eng.steps_to_episode_end() == steps and (a_eng.get_state("episode") == 2).all()
    np.testing.assert_array_equal(a_eng.get_state("record"), b_eng.get_state("record"))
    assert (a_eng.info[:, L.INFO_IDX["fault"]] == 0).all()
    print("in-kernel actor (%s): max |logit error| %.2e over %d action checks" % (activation, worst, checked))
    a_eng.close()
    b_eng.close()


@pytest.mark.parametrize("flags", [0, 1024])
def test_layernorm_of_large_relu_activations(flags):
    """ReLU outputs of ~30 with a spread of ~0.5 (large biases, small weights): the hidden LayerNorms' one-pass variance must not
    lose the spread to cancellation (advisor finding of round 3: E[x^2] - mean^2 in fp32 is ~1e-4 off at this magnitude, ten times
    eps; the moments are taken of values shifted by the first unit's)."""
    import torch
    N, K = 64, 8
    nets = [_torch_actor(300 + a, "relu") for a in range(3)]
    with torch.no_grad():
        for m in nets:
            fc = m.base.mlp.fc
            fc[0].weight.mul_(0.15); fc[0].bias.add_(30.0)
            fc[3].weight.mul_(0.15); fc[3].bias.add_(40.0)
    eng = _engine(N, 96, debug_flags=flags)
    for a in range(3):
        sd = dict(nets[a].state_dict())
        sd["activation"] = "relu"
        eng.set_actor(a, sd)
    obs0, _ = eng.reset()
    obs0 = obs0.clone()
    obs, share, rew, done, info, acts, logits = eng.rollout_actor(K, sample=False, want_logits=True)
    inp = torch.cat([obs0[None], obs[:-1]], 0)
    worst = 0.0
    for a in range(3):
        with torch.no_grad():
            x = inp[:, :, a, :].cpu()
            h1 = nets[a].base.mlp.fc[1](nets[a].base.mlp.fc[0](nets[a].base.feature_norm(x)))
            assert h1.mean() > 20 and h1.std(-1).mean() < 3        # the regime the finding is about
            ref = nets[a](x).numpy()
        worst = max(worst, float(np.abs(logits[:, :, a, :].cpu().numpy() - ref).max()))
    print("large ReLU activations: max |logit error| %.2e" % worst)
    assert worst <= 5e-4, worst
    eng.close()


def test_in_kernel_actor_sampling_follows_the_softmax():
    import torch
    N, steps = 2048, 96
    nets = [_torch_actor(7 + a) for a in range(3)]
    eng = _engine(N, steps)
    for a in range(3):
        eng.set_actor(a, nets[a].state_dict())
    eng.reset()
    obs, share, rew, done, info, acts, logits = eng.rollout_actor(48, sample=True, want_logits=True)
    p = torch.softmax(logits.double(), -1).cpu().numpy()                # [K, N, 3 agents, 3 actions]
    a = acts.cpu().numpy()
    assert a.min() >= 0 and a.max() <= 2
    for ag in range(3):
        freq = np.stack([(a[..., ag] == c).mean() for c in range(3)])
        exp = p[:, :, ag, :].mean((0, 1))
        assert np.abs(freq - exp).max() < 0.01, (ag, freq, exp)          # ~98 000 draws per agent
    # different steps / envs draw differently; the same launch twice from the same state draws the same (counter-based)
    assert len(np.unique(a[:, :, 0], axis=0)) > 1
    eng.close()


def test_rollout_actor_refuses_what_it_does_not_serve():
    eng = _engine(64, 96)
    eng.reset()
    with pytest.raises(L.SdcError, match="sdc_set_actor"):
        eng.rollout_actor(4)
    for a in range(3):
        eng.set_actor(a, _torch_actor(a).state_dict())
    with pytest.raises(L.SdcError, match="no observations yet"):
        eng.rollout_actor(4)
    eng.reset()
    with pytest.raises(L.SdcError, match="past the end"):
        eng.rollout_actor(97)
    eng.rollout_actor(4)
    eng.close()


def test_reset_keeps_the_closed_loops_observation_latch_consistent():
    """sdc_reset and the library's copy of the latest observations (what the first actions of sdc_rollout_actor are chosen
    from): a reset WITHOUT an observation buffer invalidates it (the closed loop refuses instead of acting on
    pre-reset observations), and a MASKED reset into a caller's scratch buffer takes over the masked envs' rows only."""
    import ctypes as C
    import torch
    N, steps = 64, 96
    a_eng, b_eng = _engine(N, steps, seed=9), _engine(N, steps, seed=9)
    for e in (a_eng, b_eng):
        for a in range(3):
            e.set_actor(a, _torch_actor(20 + a).state_dict())
        e.reset()
    # (a masked reset right after a full one: every env is at episode step 0, so the batch stays in lock-step -- the closed
    # loop serves lock-step batches only -- while the masked envs start a NEW episode with other draws and observations)
    mask = (np.arange(N) % 3 == 0).astype(np.uint8)
    b_eng.reset(mask=mask)                                    # the engine's own persistent buffer: the reference behaviour
    scratch = torch.full((N, 3, 26), 123.0, device="cuda")    # a C caller's scratch buffer: garbage in the unmasked rows
    share = torch.zeros((N, 29), device="cuda")
    with torch.cuda.device(a_eng.device):
        L.check(a_eng.lib.sdc_reset(a_eng._h, mask.ctypes.data_as(C.POINTER(C.c_uint8)), None, C.c_void_p(scratch.data_ptr()),
                                    C.c_void_p(share.data_ptr()), a_eng._stream()))
    assert (scratch[torch.from_numpy(mask == 0).cuda()] == 123.0).all()     # (the kernel wrote the masked rows only)
    ra = a_eng.rollout_actor(12, sample=False)
    rb = b_eng.rollout_actor(12, sample=False)
    for u, v, nm in zip(ra[:6], rb[:6], ("obs", "share", "rew", "done", "info", "actions")):
        assert torch.equal(u, v), nm
    # no observation buffer: the latch is stale, the closed loop refuses until a reset / step has delivered observations
    with torch.cuda.device(a_eng.device):
        L.check(a_eng.lib.sdc_reset(a_eng._h, None, None, None, None, a_eng._stream()))
    with pytest.raises(L.SdcError, match="no observations yet"):
        a_eng.rollout_actor(4)
    a_eng.reset()
    a_eng.rollout_actor(4)
    a_eng.close()
    b_eng.close()


def test_sampled_actions_do_not_depend_on_how_the_steps_are_launched():
    """The closed loop's sampler is keyed on (seed, global env index, episode number, episode step) -- env state and configuration
    only (round 4; round 3 keyed on the library's launch counter): the same steps asked for as 48, as 24 + 24 or as 1 + 47 draw
    the same actions and give the same trajectory."""
    import torch
    N, steps = 256, 96
    nets = [_torch_actor(70 + a) for a in range(3)]
    engs = [_engine(N, steps, seed=31) for _ in range(3)]
    for e in engs:
        for a in range(3):
            e.set_actor(a, nets[a].state_dict())
        e.reset()
    a, b, c = engs
    g = torch.Generator(device="cpu").manual_seed(3)
    ext = torch.randint(0, 3, (12, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(10):
        for e in engs:
            e.step(ext[t])
    ra = a.rollout_actor(48, sample=True)
    rb1, rb2 = b.rollout_actor(24, sample=True), b.rollout_actor(24, sample=True)
    for i, nm in enumerate(("obs", "share", "rew", "done", "info", "actions")):
        if nm == "info":
            continue
        assert torch.equal(ra[i], torch.cat([rb1[i], rb2[i]], 0)), nm
    # c: the same 48 steps as 1 + 47 after a detour through two external steps on a DIFFERENT engine state is not the same
    # trajectory -- but the same state reached by another launch pattern is: 10 steps + rollout(1) + rollout_actor(47)
    rc1 = c.rollout_actor(1, sample=True)
    rc2 = c.rollout_actor(47, sample=True)
    for i in (0, 2, 3, 5):
        assert torch.equal(ra[i], torch.cat([rc1[i], rc2[i]], 0)), i
    # a restored checkpoint has no observation latch and no feature rows for the rest of its episode: the closed loop refuses
    ck = c.state_dict()
    c.load_state_dict(ck)
    with pytest.raises(L.SdcError):
        c.rollout_actor(4, sample=True)
    for e in engs:
        e.close()
