"""Rule-based policies inside the step kernel (sdc_config.policy): the
reference's do-nothing base agents, RBCBatteryAgent and trim_and_respond_ctrl played on the device, so that
`sdc_rollout` runs whole episodes closed-loop without an action array.  Pinned by tests/golden/rbc_ny_m7.npz: the
reference env stepped by the reference's own controller classes (gen_golden.py --only rbc_ny_m7)."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from tests import gpu_helpers as G
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _run(d, mode):
    import torch
    steps, n_ep, N = int(d["meta_steps"]), int(d["meta_episodes"]), 2
    eng = G.make_engine_for_fixture(d, n_envs=N, policy=tuple(int(x) for x in d["meta_device_policy"]),
                                    trim_and_respond_limit=float(d["meta_tr_limit"]),
                                    debug_flags=1 if mode == "steps" else 0)
    eng.set_state("stpt", np.full(N, float(d["init_stpt"])))
    worst = dict(obs=0.0, rew=0.0, info=0.0)
    for ep in range(n_ep):
        pre = f"ep{ep}_"
        obs, share = eng.reset(override=G.override_from_fixture(d, ep, N, eng.lw))
        assert G.rel_err(G.raw_obs(obs.cpu().numpy())[0], d[pre + "reset_obs"]).max() <= TOL
        if mode == "rollout":      # ONE launch per episode, no action array
            O, S, R, D, I, A = eng.rollout_policy(steps)
            O, R, D, I, A = (x.cpu().numpy() for x in (O, R, D, I, A))
        else:                      # single steps through sdc_step with actions = NULL (verify mode on)
            O = np.zeros((steps, N, 3, 26), np.float32)
            R = np.zeros((steps, N, 3), np.float32)
            D = np.zeros((steps, N), np.uint8)
            I = np.zeros((steps, N, L.INFO_DIM), np.float32)
            A = None
            for t in range(steps):
                o, s, r, dn, inf = eng.step(None)
                O[t], R[t], D[t], I[t] = o.cpu().numpy(), r.cpu().numpy(), dn.cpu().numpy(), inf.cpu().numpy()
        acts = d[pre + "actions"]
        if A is not None:
            np.testing.assert_array_equal(A[:, 0], acts)           # the controllers' choices, step for step
            np.testing.assert_array_equal(A[:, 0], A[:, 1])
        # what the info block says about the choices
        np.testing.assert_array_equal(I[:, 0, L.INFO_IDX["bat_action"]], acts[:, 2])
        np.testing.assert_array_equal(I[:, 0, L.INFO_IDX["dc_crac_setpoint_delta"]], acts[:, 1] - 1)
        raw = G.raw_obs(O)
        worst["obs"] = max(worst["obs"], float(G.rel_err(raw[:, 0], d[pre + "obs"]).max()))
        worst["rew"] = max(worst["rew"], float(G.rel_err(R[:, 0], d[pre + "rew"]).max()))
        keys = [str(k) for k in d["meta_info_keys"]]
        for j, k in enumerate(keys):
            worst["info"] = max(worst["info"], float(G.rel_err(I[:, 0, L.INFO_IDX[k]], d[pre + "info"][:, j]).max()))
        np.testing.assert_array_equal(D[:, 0], d[pre + "done"])
        assert (I[:, :, L.INFO_IDX["fault"]] == 0).all()
    if mode == "steps":
        assert (eng.get_state("order_stat_sticky") == 0).all()
    eng.close()
    return worst


@pytest.mark.parametrize("mode", ["rollout", "steps"])
def test_reference_rbc_episode_closed_loop_on_device(mode):
    d = G.load_fixture("rbc_ny_m7")
    w = _run(d, mode)
    print("rbc fixture", mode, w)
    assert w["obs"] <= TOL and w["rew"] <= TOL and w["info"] <= 2e-6


def test_do_nothing_slots_and_mixed_policies():
    """A slot with POLICY_DO_NOTHING ignores the caller's column (ls 1, dc 1, bat 2: utils/base_agents.py), the others
    take it; the applied actions come back through actions_out."""
    import torch
    from dc_rl_amd import dc_config, traces
    from dc_rl_amd.engine import SdcEngine
    N, steps = 64, 96
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)

    def mk(policy):
        e = SdcEngine(N, episode_steps=steps, auto_reset=False, seed=9, policy=policy)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, 100, 110)
        return e

    g = torch.Generator(device="cpu").manual_seed(2)
    acts = torch.randint(0, 3, (steps, N, 3), dtype=torch.int32, generator=g).cuda()
    forced = acts.clone()
    forced[:, :, 0] = 1
    forced[:, :, 2] = 2
    a, b = mk((1, 0, 1)), mk((0, 0, 0))
    a.reset()
    b.reset()
    Oa, Sa, Ra, Da, Ia, Aa = a.rollout_policy(steps, acts)      # ls and bat slots ignore their columns
    Ob, Sb, Rb, Db, Ib = b.rollout(forced)
    assert torch.equal(Aa, forced)
    for x, y in ((Oa, Ob), (Ra, Rb), (Da, Db)):
        assert torch.equal(x, y)
    with pytest.raises(ValueError):
        a.rollout_policy(steps)              # agent_dc has no policy: an action array is required
    a.close()
    b.close()
    with pytest.raises(Exception):
        mk((2, 0, 0))                        # RBC is a battery policy
