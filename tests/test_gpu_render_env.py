"""`make_render_env` (harl/utils/envs_tools.py:106-133): the 5-tuple the runners unpack (on_policy_base_runner.py:120-127) and the
un-batched env their `render()` drives (:746-852) -- the loop below is that method's env-facing part, statement for statement
(reset, expand dims, step(actions[0]), rewards[0][0], the per-agent `.get` walk of the CSV dump, `if eval_dones[0]: break`)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ENV_ARGS = {"location": "ny", "month": 6, "days_per_episode": 1, "partial_obs": True,
            "nonoverlapping_shared_obs_space": True}

LS_KEYS = ['ls_original_workload', 'ls_shifted_workload', 'ls_action', 'ls_norm_load_left', 'ls_unasigned_day_load_left',
           'ls_penalty_flag', 'ls_tasks_in_queue', 'ls_tasks_dropped', 'ls_current_hour']
DC_KEYS = ['dc_ITE_total_power_kW', 'dc_HVAC_total_power_kW', 'dc_total_power_kW', 'dc_power_lb_kW', 'dc_power_ub_kW',
           'dc_crac_setpoint_delta', 'dc_crac_setpoint', 'dc_cpu_workload_fraction', 'dc_int_temperature', 'dc_CW_pump_power_kW',
           'dc_CT_pump_power_kW', 'dc_water_usage', 'dc_exterior_ambient_temp', 'outside_temp', 'day', 'hour']
BAT_KEYS = ['bat_action', 'bat_SOC', 'bat_CO2_footprint', 'bat_avg_CI', 'bat_total_energy_without_battery_KWh',
            'bat_total_energy_with_battery_KWh', 'bat_max_bat_cap', 'bat_dcload_min', 'bat_dcload_max']


def test_make_render_env_runs_the_runners_render_loop():
    from dc_rl_amd import make_render_env
    from dc_rl_amd.vec_env import SustainDCVecEnv
    args = dict(ENV_ARGS)
    env, manual_render, manual_expand_dims, manual_delay, env_num = make_render_env("sustaindc", 7, args)
    assert (manual_render, manual_expand_dims, manual_delay, env_num) == (True, True, True, 1)
    assert args["is_render"] is True and env.is_render and isinstance(env.experiment_datetime, str)
    assert env.n_agents == 3 and env.render_episode == 0
    assert [s.shape for s in env.observation_space] == [(26,)] * 3 and env.share_observation_space[0].shape == (29,)
    # the same episode on a plain one-env batch with the same seed: the render env is a re-layout of it
    vec = SustainDCVecEnv(dict(ENV_ARGS), n_envs=1, seed=7 * 60000, months=[6], auto_reset=False)
    rng = np.random.default_rng(1)
    for episode in range(2):
        eval_obs, s_obs, avail = env.reset()
        vobs, vshare, _ = vec.reset()
        assert env.render_episode == episode + 1
        assert isinstance(eval_obs, list) and len(eval_obs) == 3 and eval_obs[0].shape == (26,) and len(s_obs) == 3
        assert avail == [[1, 1, 1]] * 3
        np.testing.assert_array_equal(np.array(eval_obs), vobs[0])
        np.testing.assert_array_equal(np.array(s_obs), vshare[0])
        eval_obs = np.expand_dims(np.array(eval_obs), axis=0)
        assert eval_obs.shape == (1, 3, 26)
        rewards = 0.0
        steps = 0
        while True:
            eval_actions = rng.integers(0, 3, (1, 3, 1))          # [env_num, n_agents, 1] as the actors' outputs are stacked
            eval_obs, _, eval_rewards, eval_dones, eval_infos, avail = env.step(eval_actions[0])
            vobs, vshare, vrew, vdone, vinfos, _ = vec.step(eval_actions[:, :, 0])
            np.testing.assert_array_equal(np.array(eval_obs), vobs[0])
            np.testing.assert_array_equal(np.array(eval_rewards, dtype=np.float32), vrew[0])
            rewards += eval_rewards[0][0]
            eval_obs = np.expand_dims(np.array(eval_obs), axis=0)
            for j, keys in enumerate((LS_KEYS, DC_KEYS, BAT_KEYS)):
                row = {key: eval_infos[j].get(key, None) for key in keys}
                assert all(v is not None for v in row.values()), (j, row)
            assert eval_infos[0]["ls_action"] == int(eval_actions[0, 0, 0])
            steps += 1
            assert len(eval_dones) == 3 and all(isinstance(d, bool) for d in eval_dones)
            if eval_dones[0]:
                break
        assert steps == 96 and np.isfinite(rewards)
    env.close()
    vec.close()


def test_make_render_env_refuses_other_envs_and_fills_the_month():
    from dc_rl_amd import make_render_env
    with pytest.raises(NotImplementedError):
        make_render_env("smac", 0, {})
    args = {k: v for k, v in ENV_ARGS.items() if k != "month"}
    env = make_render_env("sustaindc", 0, args)[0]
    assert args["month"] == 0
    env.close()
