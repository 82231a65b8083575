"""The batched HARL surface against the reference's OWN HARL layer: tests/golden/harl_ny_n4.npz was
recorded through HARLSustainDCEnv (harl/envs/sustaindc/harlsustaindc_env.py, with pad_observations_v0) and
ShareDummyVecEnv (harl/envs/env_wrappers.py:301-350) for 4 envs over two auto-resets.  SustainDCVecEnv must return the
same obs [N,3,26] / share_obs [N,3,29] (slot 28 = the padded bat state's trailing zero) / rews [N,3,1] / dones [N,3] /
available actions / original_obs / original_state.
harl_ny_n2_concat.npz is the same capture with the layer's OTHER shared-observation option
(nonoverlapping_shared_obs_space False = the default of harlsustaindc_env.py:53: share_obs [N,3,78], the three padded
observations concatenated; share space Box(0, 1, (78,)), sustaindc_ptzoo.py:32-44), 2 envs over one auto-reset."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from tests import gpu_helpers as G

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("fixture,sdim,n_boundaries", [("harl_ny_n4", 29, 2), ("harl_ny_n2_concat", 78, 1)])
def test_vec_env_matches_reference_harl_layer(fixture, sdim, n_boundaries):
    from dc_rl_amd import SustainDCVecEnv
    d = G.load_fixture(fixture)
    N, steps, T = int(d["meta_n_envs"]), int(d["meta_steps"]), int(d["meta_n_steps"])
    months = [int(m) for m in d["meta_months"]]
    args = {"location": "ny", "days_per_episode": steps // 96, "partial_obs": True}
    if sdim == 29:
        args["nonoverlapping_shared_obs_space"] = True
    else:
        assert not bool(d["meta_nonoverlapping"])      # (key left out: the HARL layer's default)
    env = SustainDCVecEnv(args, n_envs=N, months=months, auto_reset=False)
    assert tuple(d["share_space_shape"]) == env.share_observation_space[0].shape == (sdim,)
    assert tuple(d["obs_space_shape"]) == env.observation_space[0].shape == (26,)
    if "share_space_low_high" in d.files:
        sp = env.share_observation_space[0]
        assert (float(np.min(sp.low)), float(np.max(sp.high))) == tuple(float(x) for x in d["share_space_low_high"])

    def share_of(obs_np, share29_np):
        """what the surface derives from a reset's raw engine outputs"""
        return obs_np.reshape(len(obs_np), 78) if sdim == 78 else share29_np
    eng = env.engine
    # the reference's tables (windows of every recorded episode at their absolute offsets) and sized constants
    W, Cc = np.zeros(L.TABLE_LEN), np.zeros(L.TABLE_LEN)
    n_ep = d["meta_episodes"]
    for i in range(N):
        for ep in range(int(n_ep[i])):
            lo = int(d[f"env{i}_ep{ep}_win_lo"])
            n = len(d[f"env{i}_ep{ep}_W"])
            W[lo:lo + n] = d[f"env{i}_ep{ep}_W"]
            Cc[lo:lo + n] = d[f"env{i}_ep{ep}_C"]
    z = np.zeros(L.TABLE_LEN)
    eng.set_tables(0, W, Cc, z, z)
    eng.set_dc_params(0, G.params_from_fixture(d))
    eng.set_state("stpt", np.full(N, float(d["init_stpt"])))
    lw = eng.lw

    def override(ep_of_env, mask):
        ov = dict(day=np.zeros(N, np.int32), hour=np.zeros(N, np.int32), ci_min=np.zeros(N), ci_max=np.ones(N),
                  t_min=np.zeros(N), t_max=np.ones(N), t_win=np.zeros((N, lw)), wb_win=np.zeros((N, lw)))
        for i in np.nonzero(mask)[0]:
            pre = f"env{i}_ep{ep_of_env[i]}_"
            c0, lo = int(d[pre + "cursor0"]), int(d[pre + "win_lo"])
            ov["day"][i], ov["hour"][i] = int(d[pre + "init_day"]), int(d[pre + "init_hour"])
            ov["ci_min"][i], ov["ci_max"][i] = float(d[pre + "ci_min30"]), float(d[pre + "ci_max30"])
            ov["t_min"][i], ov["t_max"][i] = float(d[pre + "t_min30"]), float(d[pre + "t_max30"])
            ov["t_win"][i] = d[pre + "T"][c0 - lo:c0 - lo + lw]
            ov["wb_win"][i] = d[pre + "WB"][c0 - lo:c0 - lo + lw]
        return ov

    ep = np.zeros(N, dtype=int)
    all_mask = np.ones(N, bool)
    obs, share = eng.reset(override=override(ep, all_mask))
    env._need_reset = False
    obs, share = obs.cpu().numpy(), share.cpu().numpy()
    assert np.abs(obs - d["reset_obs"]).max() <= TOL
    assert np.abs(np.repeat(share_of(obs, share)[:, None, :], 3, axis=1) - d["reset_share"]).max() <= TOL
    worst = dict(obs=0.0, share=0.0, rew=0.0)
    boundaries = 0
    for t in range(T):
        o, s, r, dn, infos, avail = env.step(d["actions"][t])
        assert o.shape == (N, 3, 26) and s.shape == (N, 3, sdim) and r.shape == (N, 3, 1) and dn.shape == (N, 3)
        np.testing.assert_array_equal(dn, d["dones"][t].astype(bool))
        np.testing.assert_array_equal(avail, d["avail"][t])
        worst["rew"] = max(worst["rew"], float(G.rel_err(r, d["rews"][t]).max()))
        done_env = dn.all(axis=1)
        if done_env.any():
            # the reference's wrapper has reset inside the call: its obs are the reset obs, the pre-reset ones sit in infos
            boundaries += 1
            for i in np.nonzero(done_env)[0]:
                worst["obs"] = max(worst["obs"], float(np.abs(infos[i][0]["original_obs"] - d["original_obs"][t, i]).max()))
                worst["share"] = max(worst["share"], float(np.abs(infos[i][0]["original_state"] - d["original_state"][t, i]).max()))
                np.testing.assert_array_equal(infos[i][0]["original_avail_actions"], np.ones((3, 3)))
                assert "original_obs" not in infos[i][1]
            ep[done_env] += 1
            ro, rs = eng.reset(mask=done_env.astype(np.uint8), override=override(ep, done_env))
            ro, rs = ro.cpu().numpy(), rs.cpu().numpy()
            worst["obs"] = max(worst["obs"], float(np.abs(ro[done_env] - d["obs"][t][done_env]).max()))
            worst["share"] = max(worst["share"], float(np.abs(share_of(ro, rs)[done_env][:, None, :] - d["share_obs"][t][done_env]).max()))
        else:
            worst["obs"] = max(worst["obs"], float(np.abs(o - d["obs"][t]).max()))
            worst["share"] = max(worst["share"], float(np.abs(s - d["share_obs"][t]).max()))
            if sdim == 29:
                assert (s[..., 28] == 0).all() and (d["share_obs"][t][..., 28] == 0).all()
            else:
                np.testing.assert_array_equal(s[:, 0], o.reshape(N, 78))     # the concatenation of the step's own observations
    print("HARL layer fixture:", worst, "auto-reset boundaries", boundaries)
    assert boundaries == n_boundaries
    assert worst["obs"] <= TOL and worst["share"] <= TOL and worst["rew"] <= TOL
    env.close()
