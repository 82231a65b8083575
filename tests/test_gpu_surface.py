"""The drop-in surface on the GPU: SustainDC (single env, dict API), SustainDCVecEnv (ShareVecEnv API with the
reference's auto-reset semantics), make_train_env."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L

pytestmark = pytest.mark.gpu

ENV_ARGS = {"location": "ny", "month": 6, "days_per_episode": 1, "partial_obs": True,
            "nonoverlapping_shared_obs_space": True}


def test_single_env_dict_api():
    from dc_rl_amd import SustainDC
    env = SustainDC(dict(ENV_ARGS), seed=3)
    assert env.agents == ["agent_ls", "agent_dc", "agent_bat"]
    assert [s.shape for s in env.observation_space] == [(26,), (14,), (13,)]
    assert [s.n for s in env.action_space] == [3, 3, 3]
    obs = env.reset()
    assert set(obs) == set(env.agents)
    assert obs["agent_ls"].shape == (26,) and obs["agent_dc"].shape == (14,) and obs["agent_bat"].shape == (13,)
    assert obs["agent_ls"].dtype == np.float32
    np.testing.assert_array_equal(obs["agent_ls"][:10], obs["agent_dc"][:10])   # shared time / CI features
    rng = np.random.default_rng(0)
    for t in range(96):
        a = {k: int(rng.integers(0, 3)) for k in env.agents}
        o, r, term, trunc, info = env.step(a)
        assert set(o) == set(env.agents) and set(r) == set(env.agents)
        assert term["__all__"] is False and trunc["__all__"] == (t == 95)
        assert all(trunc[k] == (t == 95) for k in env.agents)
        assert r["agent_dc"] == r["agent_bat"]
        c = info["__common__"]
        assert c["bat_action"] == a["agent_bat"] and c["ls_action"] == a["agent_ls"]
        assert 0.0 <= c["ls_shifted_workload"] <= 1.0 and c["dc_total_power_kW"] > 0
        assert c["bat_total_energy_with_battery_KWh"] >= 0 and c["isterminal"] == (t == 95)
        assert info["agent_ls"]["dc_crac_setpoint"] == c["dc_crac_setpoint"]
    with pytest.raises(Exception):
        env.step(a)            # episode over: the single-env surface requires reset(), like the reference
    env.reset()
    env.step(a)
    env.close()


def test_vec_env_shapes_and_auto_reset():
    from dc_rl_amd import make_train_env
    N = 24
    args = {k: v for k, v in ENV_ARGS.items() if k != "month"}
    envs = make_train_env("sustaindc", seed=5, n_threads=N, env_args=args)
    assert envs.num_envs == N and envs.n_agents == 3
    assert len(envs.observation_space) == 3 and envs.observation_space[0].shape == (26,)
    assert envs.share_observation_space[0].shape == (29,) and envs.action_space[0].n == 3
    assert envs.months[:14] == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 5, 6]
    obs, share, avail = envs.reset()
    assert obs.shape == (N, 3, 26) and share.shape == (N, 3, 29) and avail.shape == (N, 3, 3)
    assert obs.dtype == np.float32 and (avail == 1).all()
    assert (obs[:, 1, 14:] == 0).all() and (obs[:, 2, 13:] == 0).all()     # zero padding
    np.testing.assert_array_equal(share[:, 0], share[:, 2])
    np.testing.assert_array_equal(share[:, 0, :26], obs[:, 0])
    np.testing.assert_array_equal(share[:, 0, 26], obs[:, 1, 11])
    np.testing.assert_array_equal(share[:, 0, 27], obs[:, 1, 13])
    np.testing.assert_array_equal(share[:, 0, 28], obs[:, 2, 25])   # the padded bat state's last entry: 0
    day = envs.engine.get_state("day")
    from dc_rl_amd import traces
    for i, m in enumerate(envs.months):
        d0 = traces.get_init_day(m)
        assert max(0, d0 - 7) <= day[i] <= min(364, d0 + 7)
    rng = np.random.default_rng(1)
    sums = {k: 0.0 for k in ("bat_CO2_footprint", "dc_water_usage")}
    for t in range(96):
        acts = rng.integers(0, 3, size=(N, 3, 1))
        obs, share, rew, dones, infos, avail = envs.step(acts)
        assert rew.shape == (N, 3, 1) and dones.shape == (N, 3) and dones.dtype == bool
        assert len(infos) == N and len(infos[0]) == 3
        s = envs.info_sums()
        manual = sum(infos[i][0]["bat_CO2_footprint"] for i in range(N))
        assert s["bat_CO2_footprint"] == pytest.approx(manual, rel=1e-5)
        assert s["ls_unasigned_day_load_left"] == 0.0
        if t < 95:
            assert not dones.any()
    assert dones.all()
    # auto-reset inside the same call: returned obs are reset obs, originals are in infos[i][0]
    i0 = infos[0][0]
    assert i0["original_obs"].shape == (3, 26) and i0["original_state"].shape == (3, 29)
    assert i0["original_avail_actions"].shape == (3, 3)
    assert not np.array_equal(i0["original_obs"], obs[0])
    assert "original_obs" not in infos[0][1]
    assert (envs.engine.get_state("t_rel") == 0).all()
    np.testing.assert_array_equal(obs[:, 0, 10:13], 0)   # queue empty after reset: age / queue features are 0
    obs, *_ = envs.step(rng.integers(0, 3, size=(N, 3)))  # [N,3] actions accepted too; no explicit reset needed
    envs.close()


def test_vec_env_torch_outputs_stay_on_device():
    import torch
    from dc_rl_amd import SustainDCVecEnv
    envs = SustainDCVecEnv(dict(ENV_ARGS), n_envs=8, seed=1, months=[6] * 8, return_torch=True)
    obs, share, avail = envs.reset()
    assert obs.is_cuda and share.is_cuda and share.shape == (8, 3, 29)
    a = torch.randint(0, 3, (8, 3, 1), device="cuda")
    obs, share, rew, dones, infos, avail = envs.step(a)
    assert rew.is_cuda and rew.shape == (8, 3, 1) and dones.dtype == torch.bool and dones.shape == (8, 3)
    envs.close()


def test_agent_subset_uses_base_agents_and_alt_rewards():
    """A subset of agents: the others are played by the reference's base do-nothing agents (sustaindc_env.py:172-191,
    623-655); alternate reward functions by name (utils/reward_creator.py:322-334)."""
    from dc_rl_amd import SustainDC
    env = SustainDC({"agents": ["agent_ls", "agent_bat"], "location": "ny", "month": 3, "days_per_episode": 1,
                     "bat_reward": "water_usage_efficiency_reward"})
    full = SustainDC({"location": "ny", "month": 3, "days_per_episode": 1,
                      "bat_reward": "water_usage_efficiency_reward"})
    env.seed(5)
    full.seed(5)
    o = env.reset()
    of = full.reset()
    assert set(o) == {"agent_ls", "agent_bat"} and len(env.observation_space) == 2 and len(env.action_space) == 2
    rng = np.random.default_rng(0)
    for t in range(96):
        a_ls, a_bat = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        o, r, term, trunc, info = env.step({"agent_ls": a_ls, "agent_bat": a_bat})
        of, rf, _, _, info_f = full.step({"agent_ls": a_ls, "agent_dc": 1, "agent_bat": a_bat})   # hold = BaseHVACAgent
        assert set(r) == {"agent_ls", "agent_bat"} and set(o) == {"agent_ls", "agent_bat"}
        assert r["agent_ls"] == rf["agent_ls"] and r["agent_bat"] == rf["agent_bat"]
        np.testing.assert_array_equal(o["agent_ls"], of["agent_ls"])
        water = info["__common__"]["dc_water_usage"]
        assert abs(r["agent_bat"] - (-0.01 * water)) <= 1e-5 * max(1.0, abs(0.01 * water))
    assert trunc["__all__"] and trunc["agent_ls"] and "agent_dc" not in trunc
    env.close()
    full.close()


def test_env_groups_on_separate_streams_match_sequential_stepping():
    """Two engines (env groups) pinned to their own streams with use_stream(), stepped interleaved without any
    synchronisation between the groups, give exactly what the same two engines give when stepped one after the other
    on the default stream: a group's launches stay ordered on its stream, and groups share no state."""
    import torch
    import bench

    def run(pinned):
        engs = [bench.build_engine(256, 96, 0, seed=500 + g)[0] for g in range(2)]
        streams = [torch.cuda.Stream() for _ in engs]
        if pinned:
            for e, s in zip(engs, streams):
                e.use_stream(s)
        gen = torch.Generator(device="cpu").manual_seed(7)
        acts = [torch.randint(0, 3, (40, 256, 3), dtype=torch.int32, generator=gen).to("cuda:0") for _ in engs]
        torch.cuda.synchronize()
        for e in engs:
            e.reset()
        out = [[], []]
        for t in range(250):          # crosses two auto-resets (96-step episodes)
            for g, e in enumerate(engs):
                obs, share, rew, done, info = e.step(acts[g][t % 40])
                if t % 50 == 49:
                    if pinned:
                        streams[g].synchronize()
                    out[g].append((obs.clone(), rew.clone(), done.clone()))
        torch.cuda.synchronize()
        res = [[tuple(x.cpu().numpy() for x in snap) for snap in o] for o in out]
        for e in engs:
            e.close()
        return res

    a, b = run(True), run(False)
    for g in range(2):
        for sa, sb in zip(a[g], b[g]):
            for xa, xb in zip(sa, sb):
                np.testing.assert_array_equal(xa, xb)


@pytest.mark.parametrize("seed", [321, 302, 310, 327])
def test_precomputed_observation_rows_match_per_step_features(seed):
    """The trace-only observation entries come from rows computed once per episode (sdc_features.hip, one lane per
    step); after a host write to the env's state the step computes them itself (all lanes of the wavefront on one
    step).  Both follow the reference's arithmetic IN THE SAME ORDER: same observations, to the bit, over episodes that
    include the start-of-year cursor edge (no past CI window) and device-side auto-resets.  (Seeds 302, 310, 327: weather draws
    under which a temperature window is clipped flat and its least-squares slope is ~1e-20 -- there the whole-wavefront
    path's butterfly sums of rounds 1-3 showed in the fp32 observation; round 4 sums in lane order, sdc_device.hpp
    seg3_seq_sum_f64.)"""
    import torch
    import bench

    def run(fallback):
        eng = bench.build_engine(192, 96, 0, seed=seed)[0]
        gen = torch.Generator(device="cpu").manual_seed(11)
        acts = torch.randint(0, 3, (64, 192, 3), dtype=torch.int32, generator=gen).to("cuda:0")
        eng.reset()
        snaps = []
        for t in range(300):
            if fallback and t % 96 == 0:
                eng.set_state("episode", eng.get_state("episode"))   # a host write: the rows count as stale
            obs, share, rew, done, info = eng.step(acts[t % 64])
            snaps.append((obs.cpu().numpy().copy(), share.cpu().numpy().copy(), rew.cpu().numpy().copy()))
        eng.close()
        return snaps

    a, b = run(False), run(True)
    for t, (sa, sb) in enumerate(zip(a, b)):
        for xa, xb in zip(sa, sb):
            assert np.array_equal(xa, xb), (t, np.abs(xa - xb).max())


def test_rollout_matches_single_steps():
    """sdc_rollout (K env-steps per launch, every wavefront advancing its own env K times) gives what K calls of
    sdc_step give, to the bit -- across window re-centrings, an auto-reset at the end of a rollout, and the
    episode-boundary guard."""
    import torch
    import bench
    from dc_rl_amd._lib import SdcError

    N, EP = 256, 96
    gen = torch.Generator(device="cpu").manual_seed(3)
    acts = torch.randint(0, 3, (3 * EP, N, 3), dtype=torch.int32, generator=gen).to("cuda:0")
    a = bench.build_engine(N, EP, 0, seed=77)[0]
    b = bench.build_engine(N, EP, 0, seed=77)[0]
    a.reset()
    b.reset()
    t = 0
    for K in (1, 7, 40, 48, 30, 66, 96):          # 48 ends the first episode, 66 the second, 96 is a whole one
        assert b.steps_to_episode_end() >= K
        obs, share, rew, done, info = b.rollout(acts[t:t + K])
        for k in range(K):
            o1, s1, r1, d1, i1 = a.step(acts[t + k])
            np.testing.assert_array_equal(o1.cpu().numpy(), obs[k].cpu().numpy())
            np.testing.assert_array_equal(s1.cpu().numpy(), share[k].cpu().numpy())
            np.testing.assert_array_equal(r1.cpu().numpy(), rew[k].cpu().numpy())
            np.testing.assert_array_equal(d1.cpu().numpy(), done[k].cpu().numpy())
            ia, ib = i1.cpu().numpy().copy(), info[k].cpu().numpy().copy()
            ia[:, 39] = ib[:, 39] = 0      # the one scheduling-dependent column: WHO re-centred a rank window (diagnostic)
            np.testing.assert_array_equal(ia, ib)
        np.testing.assert_array_equal(a.final_obs.cpu().numpy(), b.final_obs.cpu().numpy())
        t += K
    assert b.steps_to_episode_end() == EP
    with pytest.raises(SdcError):
        b.rollout(acts[:EP + 1])                   # would run past the end of the episode
    a.close()
    b.close()


def test_vec_env_agent_subset_and_device_logger_sums():
    """Two trained agents: agent_dc is played by the base do-nothing agent ON THE DEVICE (policy slot), the surface
    carries two agents; the logger sums accumulate on the device and are read once."""
    from dc_rl_amd import SustainDCVecEnv
    N = 16
    args = dict(ENV_ARGS, agents=["agent_ls", "agent_bat"])
    sub = SustainDCVecEnv(args, n_envs=N, seed=3, months=[6] * N)
    ref = SustainDCVecEnv(dict(ENV_ARGS), n_envs=N, seed=3, months=[6] * N)
    assert sub.n_agents == 2 and sub.agents == ["agent_ls", "agent_bat"] and sub.policy == (0, 1, 0)
    assert len(sub.observation_space) == len(sub.action_space) == len(sub.share_observation_space) == 2
    o1, s1, a1 = sub.reset()
    o3, s3, a3 = ref.reset()
    assert o1.shape == (N, 2, 26) and s1.shape == (N, 2, 29) and a1.shape == (N, 2, 3)
    np.testing.assert_array_equal(o1, o3[:, [0, 2]])
    ref.accumulate_logger_sums()
    rng = np.random.default_rng(5)
    manual = {k: 0.0 for k in ("bat_CO2_footprint", "dc_water_usage", "ls_tasks_in_queue")}
    kept = None
    for t in range(40):
        a2 = rng.integers(0, 3, size=(N, 2, 1))
        full = np.ones((N, 3, 1), dtype=np.int64)
        full[:, [0, 2]] = a2
        o1, s1, r1, d1, i1, _ = sub.step(a2)
        o3, s3, r3, d3, i3, _ = ref.step(full)          # the same thing spelled out: agent_dc always holds (1)
        assert r1.shape == (N, 2, 1) and d1.shape == (N, 2)
        np.testing.assert_array_equal(o1, o3[:, [0, 2]])
        np.testing.assert_array_equal(r1, r3[:, [0, 2]])
        np.testing.assert_array_equal(s1[:, 0], s3[:, 0])
        assert i1[3][0]["dc_crac_setpoint_delta"] == 0.0
        for k in manual:
            manual[k] += sum(i3[i][0][k] for i in range(N))
        if t == 10:
            kept = (i3, float(i3[2][0]["bat_SOC"]), float(i3[5][0]["dc_total_power_kW"]))
    # an infos object read later still shows ITS step (snapshot), not the latest one
    assert float(kept[0][2][0]["bat_SOC"]) == kept[1] and float(kept[0][5][0]["dc_total_power_kW"]) == kept[2]
    # ... and one that was never touched before its info block was overwritten raises instead of showing a later step
    a2 = rng.integers(0, 3, size=(N, 2, 1))
    stale = sub.step(a2)[4]
    fresh = sub.step(a2)[4]
    _ = sub.step(a2)
    assert fresh[0][0]["bat_SOC"] >= 0.0            # NumPy mode: the pinned double buffer holds one more step
    with pytest.raises(RuntimeError, match="earlier step"):
        stale[0][0]["bat_SOC"]
    sums, n = ref.read_logger_sums()
    assert n == 40
    for k, v in manual.items():
        assert sums[k] == pytest.approx(v, rel=1e-5)
    assert sums["ls_unasigned_day_load_left"] == 0.0
    sub.close()
    ref.close()


def test_last_done_host_mirror_matches_device_done():
    """sdc_last_done: the `done` output from the host's mirror of the step counters (no device read) -- what lets the
    device-resident ShareVecEnv path run without a host synchronisation per step.  Staggered by a masked reset."""
    import torch
    import bench
    N, EP = 40, 24
    eng = bench.build_engine(N, EP, 0, seed=5)[0]
    eng.reset()
    g = torch.Generator(device="cpu").manual_seed(0)
    acts = torch.randint(0, 3, (80, N, 3), dtype=torch.int32, generator=g).cuda()
    seen = 0
    for t in range(80):
        if t == 7:
            eng.reset(mask=(np.arange(N) % 4 == 1).astype(np.uint8))     # a quarter of the envs restart: two phases
        obs, share, rew, done, info = eng.step(acts[t])
        d = done.cpu().numpy().astype(bool)
        ld = eng.last_done()
        if d.any():
            seen += 1
            np.testing.assert_array_equal(ld, d)
        else:
            assert ld is None
    assert seen >= 5
    # ... and after a rollout that ends an episode
    k = eng.steps_to_episode_end()
    o, s, r, dn, i = eng.rollout(acts[:k].contiguous())
    np.testing.assert_array_equal(eng.last_done(), dn[-1].cpu().numpy().astype(bool))
    eng.close()


@pytest.mark.parametrize("agents,width", [(["agent_ls", "agent_bat"], 26), (["agent_dc", "agent_bat"], 14)])
def test_agent_subset_concat_share_obs_on_the_torch_path(agents, width):
    """An agent SUBSET with the HARL layer's concatenated shared observation (nonoverlapping_shared_obs_space False) and
    device-resident outputs: `share_obs` is the trained agents' padded observations of THIS step, every step (round 3 cached
    the first step's copy: ADVICE r3), padded to the widest TRAINED agent like ss.pad_observations_v0
    (harlsustaindc_env.py:25-26; sustaindc_ptzoo.py:32-44: dc + bat -> 2 x 14)."""
    import torch
    from dc_rl_amd import SustainDCVecEnv
    N = 16
    args = dict(ENV_ARGS, agents=agents, nonoverlapping_shared_obs_space=False)
    env = SustainDCVecEnv(args, n_envs=N, seed=3, months=[6] * N, return_torch=True)
    ref = SustainDCVecEnv(args, n_envs=N, seed=3, months=[6] * N)                      # NumPy outputs, same seed
    k = len(agents)
    assert env.observation_space[0].shape == (width,) and env.share_observation_space[0].shape == (width * k,)
    o, s, _ = env.reset()
    on, sn, _ = ref.reset()
    assert o.shape == (N, k, width) and s.shape == (N, k, width * k)
    g = torch.Generator(device="cpu").manual_seed(2)
    prev = None
    for t in range(100):                                   # one auto-reset inside (96-step episodes)
        a = torch.randint(0, 3, (N, k, 1), generator=g)
        o, s, r, d, infos, _ = env.step(a.cuda())
        on, sn, rn, dn, infn, _ = ref.step(a.numpy())
        assert torch.equal(s[:, 0], o.reshape(N, -1)) and torch.equal(s[:, 0], s[:, k - 1]), t
        np.testing.assert_array_equal(o.cpu().numpy(), on)
        np.testing.assert_array_equal(s.cpu().numpy(), sn)
        if prev is not None:
            assert not torch.equal(prev, s), t             # (it moves: not the first step's values)
        prev = s.clone()
        if dn.all():
            fo = infn[3][0]["original_obs"]
            assert fo.shape == (k, width) and infn[3][0]["original_state"].shape == (k, width * k)
            np.testing.assert_array_equal(infos[3][0]["original_obs"], fo)
    env.close()
    ref.close()


def test_eval_dump_keys_of_the_reference_runner():
    """The evaluation dump of the reference's runner reads 34 keys per env and step through `eval_infos[i][j].get(key, None)`
    (harl/runners/on_policy_base_runner.py:617-638: 9 for agent 1, 16 for agent 2, 9 for agent 3): every one of them is served
    (none falls back to the `None` default), with the values of the step's info block / the env's constants."""
    from dc_rl_amd import make_eval_env
    keys = {0: ['ls_original_workload', 'ls_shifted_workload', 'ls_action', 'ls_norm_load_left', 'ls_unasigned_day_load_left',
                'ls_penalty_flag', 'ls_tasks_in_queue', 'ls_tasks_dropped', 'ls_current_hour'],
            1: ['dc_ITE_total_power_kW', 'dc_HVAC_total_power_kW', 'dc_total_power_kW', 'dc_power_lb_kW', 'dc_power_ub_kW',
                'dc_crac_setpoint_delta', 'dc_crac_setpoint', 'dc_cpu_workload_fraction', 'dc_int_temperature',
                'dc_CW_pump_power_kW', 'dc_CT_pump_power_kW', 'dc_water_usage', 'dc_exterior_ambient_temp', 'outside_temp', 'day',
                'hour'],
            2: ['bat_action', 'bat_SOC', 'bat_CO2_footprint', 'bat_avg_CI', 'bat_total_energy_without_battery_KWh',
                'bat_total_energy_with_battery_KWh', 'bat_max_bat_cap', 'bat_dcload_min', 'bat_dcload_max']}
    assert sum(len(v) for v in keys.values()) == 34
    N = 6
    envs = make_eval_env("sustaindc", seed=2, n_threads=N, env_args=dict(ENV_ARGS))
    envs.reset()
    rng = np.random.default_rng(0)
    for t in range(5):
        a = rng.integers(0, 3, size=(N, 3, 1))
        _, _, _, _, infos, _ = envs.step(a)
        rows = infos.rows()
        for i in range(N):
            for j in range(3):
                got = {k: infos[i][j].get(k, None) for k in keys[j]}
                assert all(v is not None for v in got.values()), (i, j, [k for k, v in got.items() if v is None])
            assert infos[i][0].get('ls_action') == int(a[i, 0, 0]) and infos[i][2].get('bat_action') == float(a[i, 2, 0])
            assert infos[i][1].get('dc_total_power_kW') == float(rows[i, L.INFO_IDX['dc_total_power_kW']]) > 0
            assert infos[i][1].get('dc_power_ub_kW') > infos[i][1].get('dc_power_lb_kW') > 0
            assert infos[i][2].get('bat_dcload_max') == infos[i][1].get('dc_power_ub_kW') / 4
    envs.close()
