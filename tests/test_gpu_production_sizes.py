"""Oracle parity in the production configuration at the sizes / kernels the bench line quotes rates for:

  * the four-envs-per-wavefront step kernel (`sdc_dynamics_quad_kernel`: the default above 5 632 envs) at 12 288 envs
    (three whole occupancy rounds), 16 384 envs (a fourth wavefront per SIMD) and 32 768 envs (eight per SIMD, three resident:
    the largest batch a rate is quoted for), against the ORACLE --
    until now it was only compared with the two-env kernels, and never above 8 200 envs;
  * BASELINE configs[3] at its own size: 4096 envs x 16 / 20 / 25 racks x three locations with `debug_flags = 0`, full rings
    and deferred re-centring under the full request load (round 4: served by the common-case kernels, every env carrying
    its own copy of its config's scalars; their bit-equality with the general kernels on such a batch:
    tests/test_gpu_timed_config.py::test_common_case_kernels_serve_several_configs);
  * `sdc_rollout` (48 steps in one launch) and the closed loop `sdc_rollout_actor` at 16 384 envs with full rings (the
    four-env multi-step kernels): sampled envs against the oracle, the closed loop under the actions its actors chose.

The recipe is tests/production_rig.py (shared with test_timed_configuration_4096_envs_vs_oracle).
Reference: sustaindc_env.py:533-621, utils/reward_creator.py:16-45."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from tests.production_rig import ProductionRig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [6144, 8192, 12288, 16384, 32768])
def test_default_step_kernel_of_large_batches_production_vs_oracle(N):
    """The default kernel of a large batch (debug_flags = 0) -- four envs per wavefront at 6 144 envs, ONE LANE PER ENV (sdc_wide.hip)
    from 7 680 -- at the sizes rates are quoted for up to 32 768 envs: 330 single steps over two auto-resets, the first / last
    wavefronts and both sides of every occupancy round sampled, every reward-state path."""
    rig = ProductionRig(N, debug_flags=0, episode_steps=120, seed=1000 + N, envs_per_wave=4)
    assert len(rig.sample) >= 72
    obs, _ = rig.eng.reset()
    rig.begin_all(obs)
    rig.single_steps(330)
    print(f"default step kernel, {N} envs:", rig.eng.last_step_kernel(), rig.worst, "reward-state paths:", rig.paths[:4],
          "auto-resets:", rig.resets, "sampled envs:", len(rig.sample))
    assert rig.resets >= 2
    assert rig.eng.last_step_kernel() == ("sdc_dynamics_quad_kernel" if N < 7680 else "sdc_dynamics_wide_kernel")
    rig.assert_ok()
    rig.assert_all_reward_state_paths_seen()
    rig.eng.close()


@pytest.mark.parametrize("N", [16384, 32768, 49152])      # (49 152: the smallest batch with the ring's slot-major mirror, SdcDev::hist_t)
def test_lane_per_env_kernel_equals_four_per_wavefront_for_every_env(N):
    """The oracle is run for a sample of the envs; EVERY env of the lane-per-env kernel is held to the four-envs-per-wavefront kernel
    here: two engines in the production configuration (full rings with duplicates, spread write positions, the same seed), 16 384
    and 32 768 envs -- and 49 152, where the lane-per-env kernel reads the key a step evicts from the ring's slot-major mirror and the
    other kernel from the ring --, 260 single steps over two auto-resets under the full load of deferred re-centrings -- every output
    of every env the same bits (the diagnostics column aside: it says which path served the reward state), and at the end the rings."""
    import torch
    a = ProductionRig(N, debug_flags=0, episode_steps=120, seed=515, envs_per_wave=4, n_random=0)
    b = ProductionRig(N, debug_flags=4096, episode_steps=120, seed=515, envs_per_wave=4, n_random=0)
    a.eng.reset()
    b.eng.reset()
    g = torch.Generator(device="cpu").manual_seed(515)
    rsv = L.INFO_IDX["reserved"]
    for t in range(260):
        acts = torch.randint(0, 3, (N, 3), dtype=torch.int32, generator=g).cuda()
        for u, v, nm in zip(a.eng.step(acts), b.eng.step(acts), ("obs", "share_obs", "rew", "done", "info")):
            if nm == "info":
                u, v = u.clone(), v.clone()
                u[:, rsv] = 0
                v[:, rsv] = 0
            if not torch.equal(u, v):
                bad = (u != v).nonzero()
                raise AssertionError((t, nm, bad[:6].tolist(), u[tuple(bad[0])].item(), v[tuple(bad[0])].item()))
    assert a.eng.last_step_kernel() == "sdc_dynamics_wide_kernel" and b.eng.last_step_kernel() == "sdc_dynamics_quad_kernel"
    if N >= 49152:
        import numpy as np
        np.testing.assert_array_equal(a.eng.get_state("hist").view(np.uint32), b.eng.get_state("hist").view(np.uint32))
    for r in (a, b):
        assert (r.eng.info[:, L.INFO_IDX["fault"]] == 0).all()
        r.eng.close()


def test_quad_step_kernel_32768_envs_vs_oracle():
    """... and the four-envs-per-wavefront kernel at 32 768 envs (debug_flags bit 12 keeps the lane-per-env kernel off): the mapping
    the closed loop (and `sdc_rollout` with `actions_out` / unaligned outputs) still runs at that size."""
    rig = ProductionRig(32768, debug_flags=4096, episode_steps=120, seed=33768, envs_per_wave=4)
    obs, _ = rig.eng.reset()
    rig.begin_all(obs)
    rig.single_steps(150)
    assert rig.eng.last_step_kernel() == "sdc_dynamics_quad_kernel" and rig.resets >= 1
    rig.assert_ok()
    rig.eng.close()


def test_65536_envs_production_vs_oracle():
    """The largest batch a rate is quoted for (`secondary.batch_scan`): 65 536 envs on the lane-per-env kernel, production
    configuration, 150 single steps over an auto-reset; first / last wavefronts, both sides of every occupancy round and a random
    spread sampled against the oracle."""
    rig = ProductionRig(65536, debug_flags=0, episode_steps=120, seed=65536, envs_per_wave=4, n_random=40)
    obs, _ = rig.eng.reset()
    rig.begin_all(obs)
    rig.single_steps(150)
    print("65536 envs:", rig.worst, "reward-state paths:", rig.paths[:4], "sampled envs:", len(rig.sample))
    assert rig.eng.last_step_kernel() == "sdc_dynamics_wide_kernel" and rig.resets >= 1
    rig.assert_ok()
    rig.assert_all_reward_state_paths_seen()
    rig.eng.close()


def test_config3_mixed_racks_32768_production():
    """BASELINE configs[3] at the size of configs[4]: 32 768 envs, rack count 20 / 16 / 25 by env_id % 3, three locations -- nine
    configs in every wavefront -- on the lane-per-env kernel's general form (what `secondary.mixed_racks_32768` times), full rings,
    debug_flags = 0, 330 steps over two auto-resets vs the oracle."""
    rig = ProductionRig(32768, debug_flags=0, mixed=True, episode_steps=120, seed=3303, envs_per_wave=4)
    combos = {(int(rig.loc_id[i]), int(rig.cfg_id[i])) for i in rig.sample}
    assert len(combos) == 9, combos
    obs, _ = rig.eng.reset()
    rig.begin_all(obs)
    rig.single_steps(330)
    print("mixed racks, 32768 envs:", rig.worst, "reward-state paths:", rig.paths[:4], "auto-resets:", rig.resets)
    assert rig.eng.last_step_kernel() == "sdc_dynamics_wide_gen_kernel" and rig.resets >= 2
    rig.assert_ok()
    rig.assert_all_reward_state_paths_seen()
    rig.eng.close()


def test_policy_and_tou_rollouts_16384_on_the_general_form_vs_oracle():
    """`sdc_rollout` with the rule-based policies (do-nothing ls agent, trim-and-respond on the CRAC set-point, RBCBatteryAgent) and
    `tou_reward` / `water_usage_efficiency_reward` for the dc / battery agents, 16 384 envs x the configs[3] mix, full rings: K
    launches of the lane-per-env kernel's general form per call, no action array.  The oracle steps the sampled envs under the
    actions the policies chose (`actions_out`); the choices themselves are checked for EVERY env against the rules
    (utils/rbc_agents.py:21-47: charge when the carbon intensity three steps ahead is above the current one;
    utils/trim_and_respond.py:28-38 on the room temperature the previous step reported)."""
    import torch
    N, limit = 16384, 34.9
    rig = ProductionRig(N, debug_flags=0, mixed=True, episode_steps=120, seed=1616, envs_per_wave=4, reward_method=(0, 3, 6),
                        policy=(1, 3, 2), trim_and_respond_limit=limit)
    eng = rig.eng
    obs, _ = eng.reset()
    rig.begin_all(obs)
    room_prev = np.full(N, np.nan)          # dc_int_temperature of the previous step (the first step of a run reads the record's)
    counter = np.zeros(N, np.int64)
    ROOM = L.INFO_IDX["dc_int_temperature"]
    n_calls = 0
    while rig.resets < 1 or n_calls < 4:
        k = min(40, eng.steps_to_episode_end())
        cur = eng.get_state("cursor")
        loc = rig.loc_id
        out = eng.rollout_policy(k)
        n_calls += 1
        assert eng.last_step_kernel() == "sdc_dynamics_wide_gen_kernel"
        A = out[5].cpu().numpy()
        info = out[4].cpu().numpy()
        assert (A[:, :, 0] == 1).all()
        C = np.stack([rig.tables[li]["C"] for li in range(len(rig.tables))])
        for t in range(k):
            i = cur + t
            want_bat = np.where(C[loc, np.minimum(i + 3, C.shape[1] - 1)] > C[loc, i], 0, 1)
            assert (A[t, :, 2] == want_bat).mean() > 0.999      # (ties of the NORMALISED values aside)
            # (the controller reads the fp64 room temperature of the record, the test the fp32 info column: envs within 1e-3 of the
            # limit are not judged; neither is the very first step, whose "previous" temperature is the record's initial 0)
            sure = ~np.isnan(room_prev) & (np.abs(room_prev - limit) > 1e-3)
            want_dc = np.where(limit >= room_prev, np.where(counter > 4, 2, 1), 0)
            assert (A[t, sure, 1] == want_dc[sure]).all()
            counter = np.where(A[t, :, 1] == 2, 0, np.where(A[t, :, 1] == 1, counter + 1, counter))
            room_prev = info[t, :, ROOM].astype(np.float64)
        rig.check_rollout(torch.from_numpy(A), out)
        if eng.steps_to_episode_end() == rig.steps:
            rig.resets += 1
            rig.begin_all(out[0][-1])
    print("policy rollouts, 16384 envs:", rig.worst, "reward-state paths:", rig.paths[:4], "calls:", n_calls)
    assert len(np.unique(A[:, :, 1])) >= 2
    rig.assert_ok()
    rig.eng.close()


def test_config3_mixed_racks_4096_production():
    """BASELINE configs[3]: 4096 envs, rack count 20 / 16 / 25 by env_id % 3, three locations, debug_flags = 0 (what the bench's
    `secondary.mixed_racks` times), full rings, 330 steps over two auto-resets vs the oracle."""
    rig = ProductionRig(4096, debug_flags=0, mixed=True, episode_steps=120, seed=303, envs_per_wave=2)
    # every (location, rack count) combination is in the sample
    combos = {(int(rig.loc_id[i]), int(rig.cfg_id[i])) for i in rig.sample}
    assert len(combos) == 9, combos
    obs, _ = rig.eng.reset()
    rig.begin_all(obs)
    rig.single_steps(330)
    print("mixed racks, 4096 envs:", rig.worst, "reward-state paths:", rig.paths[:4], "auto-resets:", rig.resets)
    assert rig.resets >= 2
    rig.assert_ok()
    rig.assert_all_reward_state_paths_seen()
    rig.eng.close()


@pytest.mark.parametrize("flags,kernel", [(0, "sdc_dynamics_wide_kernel"), (4096, "sdc_rollout_quad_kernel")])
def test_rollout_16384_envs_full_rings_vs_oracle(flags, kernel):
    """`sdc_rollout` at 16 384 envs, rings full: 10 single steps (their deferred requests are still in flight when the multi-step
    call starts), then 48 + 48 + the episode's last 14 steps in three calls, across an auto-reset, vs the oracle.  debug_flags 0:
    what the call does at this size since round 5 -- K single-step launches of the lane-per-env kernel; bit 12 (4096) keeps that
    kernel off: ONE launch of `sdc_rollout_quad_kernel` per call (still what serves unaligned outputs / `actions_out`)."""
    import torch
    N = 16384
    rig = ProductionRig(N, debug_flags=flags, episode_steps=120, seed=555, envs_per_wave=4)
    eng = rig.eng
    obs, _ = eng.reset()
    rig.begin_all(obs)
    rig.single_steps(10)
    g = torch.Generator(device="cpu").manual_seed(91)
    while rig.resets < 1:
        k = min(48, eng.steps_to_episode_end())
        acts = torch.randint(0, 3, (k, N, 3), dtype=torch.int32, generator=g).cuda()
        out = eng.rollout(acts)
        rig.check_rollout(acts, out)
        if eng.steps_to_episode_end() == rig.steps:
            rig.resets += 1
            rig.begin_all(out[0][-1])
    k = 24
    acts = torch.randint(0, 3, (k, N, 3), dtype=torch.int32, generator=g).cuda()
    rig.check_rollout(acts, eng.rollout(acts))
    print("sdc_rollout, 16384 envs:", eng.last_step_kernel(), rig.worst, "reward-state paths:", rig.paths[:4])
    assert eng.last_step_kernel() == kernel
    rig.assert_ok()
    rig.eng.close()


def test_rollout_actor_16384_envs_full_rings_vs_oracle():
    """The closed loop at 16 384 envs (sdc_rollout_actor_quad_kernel), rings full: two 48-step launches with sampled actions;
    the oracle steps the sampled envs under the actions the in-kernel actors chose (`actions_out`)."""
    from tests.test_gpu_actor import _torch_actor
    N = 16384
    rig = ProductionRig(N, debug_flags=0, episode_steps=120, seed=556, envs_per_wave=4)
    eng = rig.eng
    for a in range(3):
        eng.set_actor(a, _torch_actor(60 + a, "tanh").state_dict())
    obs, _ = eng.reset()
    rig.begin_all(obs)
    rig.single_steps(6)
    seen = np.zeros(3, np.int64)
    for sample in (True, False):
        out = eng.rollout_actor(48, sample=sample)
        acts = out[5]
        a = acts.cpu().numpy()
        assert a.min() >= 0 and a.max() <= 2
        seen += np.bincount(a.reshape(-1), minlength=3)
        rig.check_rollout(acts, out)
    assert (seen > 0).all(), seen       # the networks do use all three actions (the oracle saw every branch)
    print("sdc_rollout_actor, 16384 envs:", rig.worst, "actions chosen:", seen.tolist())
    rig.assert_ok()
    assert (eng.info[:, L.INFO_IDX["fault"]] == 0).all()
    rig.eng.close()


def test_real_episode_length_4096_production_vs_oracle():
    """The bench's own episode shape -- 4096 envs, 672-step (7-day) episodes, debug_flags 0, rings full -- over two whole
    episodes and into a third (1400 steps, two auto-resets 672 steps apart): the long horizon the 120-step rigs do not reach
    (a rank window re-centred ~every 600 steps per env: every sampled env goes through several deferred take-overs)."""
    rig = ProductionRig(4096, debug_flags=0, episode_steps=672, seed=909, envs_per_wave=2, n_random=40)
    obs, _ = rig.eng.reset()
    rig.begin_all(obs)
    rig.single_steps(1400)
    print("672-step episodes, 4096 envs:", rig.worst, "reward-state paths:", rig.paths[:4], "auto-resets:", rig.resets)
    assert rig.resets == 2
    rig.assert_ok()
    rig.assert_all_reward_state_paths_seen()
    assert rig.paths[2] > 10 * len(rig.sample) // 72     # deferred take-overs did happen, many times over
    rig.eng.close()
