#!/usr/bin/env python3
"""What an UNCHANGED single-process HARL runner gets out of the device batch: the runner's per-step access pattern restated
loop for loop around `SustainDCVecEnv.step` (NumPy in, NumPy out -- the reference's runner keeps its buffers on the host).

Restated (nothing imported from the reference):
  * `OnPolicyBaseRunner.run` (harl/runners/on_policy_base_runner.py:259-282): actions [N, 3, 1] from the actors (here: drawn
    with NumPy -- the networks' cost is the trainer's, not the env's; leaving it out makes the env-side share LARGER),
    `envs.step(actions)`, `logger.per_step(data)`, `insert(data)`;
  * `SustainDCLogger.per_step` (harl/envs/sustaindc/sustaindc_logger.py:80-101): for every env `infos[i][0].get(key, 0)` for
    the ten metric keys, the `> 0` test with its second and third read of dc_HVAC_total_power_kW, the step counter;
  * `insert` (on_policy_base_runner.py:386-503): `dones_env`, the rnn-state / mask resets, `bad_masks` through
    `"bad_transition" in info[0].keys()` for every env, the actor / critic buffer inserts (array copies into [T + 1, N, ...]).

Reported per batch size: us per runner step, env-steps/s, the time inside `envs.step`, the time of the SAME walk over plain
pre-built dicts (the runner's own Python: what it would cost with the reference's materialised infos, transport excluded),
and the env-side share = (whole loop - the runner's own Python) / whole loop.
usage: python tools/harl_loop_rate.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LOGGER_METRICS = [("net_energy_sum", "bat_total_energy_with_battery_KWh"), ("CO2_footprint_sum", "bat_CO2_footprint"),
                  ("water_usage", "dc_water_usage"), ("load_left", "ls_unasigned_day_load_left"),
                  ("ls_tasks_in_queue", "ls_tasks_in_queue"), ("ls_tasks_dropped", "ls_tasks_dropped"),
                  ("ite_power_sum", "dc_ITE_total_power_kW"), ("ct_power_sum", "dc_CT_total_power_kW"),
                  ("chiller_power_sum", "dc_Compressor_total_power_kW"), ("hvac_power_sum", "dc_HVAC_total_power_kW")]


class RunnerSide:
    """The host-side state an on-policy runner keeps per step (episode_length T, N threads, 3 agents, hidden 64)."""

    def __init__(self, N, T, obs_dim, share_dim, n_agents=3, hidden=64):
        self.N, self.T, self.k, self.h = N, T, n_agents, hidden
        self.metrics = {m: 0 for m, _ in LOGGER_METRICS}
        self.metrics.update(hvac_power_on_used=[], step_count=0)
        self.obs = [np.zeros((T + 1, N, obs_dim), np.float32) for _ in range(n_agents)]
        self.rnn = [np.zeros((T + 1, N, 1, hidden), np.float32) for _ in range(n_agents)]
        self.act = [np.zeros((T, N, 1), np.float32) for _ in range(n_agents)]
        self.logp = [np.zeros((T, N, 1), np.float32) for _ in range(n_agents)]
        self.masks = [np.ones((T + 1, N, 1), np.float32) for _ in range(n_agents)]
        self.active = [np.ones((T + 1, N, 1), np.float32) for _ in range(n_agents)]
        self.share = np.zeros((T + 1, N, share_dim), np.float32)
        self.rnn_c = np.zeros((T + 1, N, 1, hidden), np.float32)
        self.values = np.zeros((T + 1, N, 1), np.float32)
        self.rewards = np.zeros((T, N, 1), np.float32)
        self.cmasks = np.ones((T + 1, N, 1), np.float32)
        self.bad = np.ones((T + 1, N, 1), np.float32)
        self.t = 0

    def per_step(self, infos):                       # sustaindc_logger.py:80-101
        m = self.metrics
        for i in range(len(infos)):
            m["net_energy_sum"] += infos[i][0].get("bat_total_energy_with_battery_KWh", 0)
            m["CO2_footprint_sum"] += infos[i][0].get("bat_CO2_footprint", 0)
            m["water_usage"] += infos[i][0].get("dc_water_usage", 0)
            m["load_left"] += infos[i][0].get("ls_unasigned_day_load_left", 0)
            m["ls_tasks_in_queue"] += infos[i][0].get("ls_tasks_in_queue", 0)
            m["ls_tasks_dropped"] += infos[i][0].get("ls_tasks_dropped", 0)
            m["ite_power_sum"] += infos[i][0].get("dc_ITE_total_power_kW", 0)
            m["ct_power_sum"] += infos[i][0].get("dc_CT_total_power_kW", 0)
            m["chiller_power_sum"] += infos[i][0].get("dc_Compressor_total_power_kW", 0)
            m["hvac_power_sum"] += infos[i][0].get("dc_HVAC_total_power_kW", 0)
            if infos[i][0].get("dc_HVAC_total_power_kW", 0) > 0:
                m["hvac_power_on_used"].append(infos[i][0].get("dc_HVAC_total_power_kW", 0))
            m["step_count"] += 1

    def insert(self, obs, share_obs, rewards, dones, infos, values, actions, logp, rnn_states, rnn_states_critic):
        N, k, h = self.N, self.k, self.h              # on_policy_base_runner.py:386-503 (state_type "EP")
        dones_env = np.all(dones, axis=1)
        rnn_states[dones_env == True] = np.zeros(((dones_env == True).sum(), k, 1, h), dtype=np.float32)
        rnn_states_critic[dones_env == True] = np.zeros(((dones_env == True).sum(), 1, h), dtype=np.float32)
        masks = np.ones((N, k, 1), dtype=np.float32)
        masks[dones_env == True] = np.zeros(((dones_env == True).sum(), k, 1), dtype=np.float32)
        active_masks = np.ones((N, k, 1), dtype=np.float32)
        active_masks[dones == True] = np.zeros(((dones == True).sum(), 1), dtype=np.float32)
        active_masks[dones_env == True] = np.ones(((dones_env == True).sum(), k, 1), dtype=np.float32)
        bad_masks = np.array([[0.0] if "bad_transition" in info[0].keys() and info[0]["bad_transition"] == True else [1.0]
                              for info in infos])
        t = self.t
        for a in range(k):
            self.obs[a][t + 1] = np.stack(obs[:, a], axis=0)
            self.rnn[a][t + 1] = rnn_states[:, a]
            self.act[a][t] = actions[:, a]
            self.logp[a][t] = logp[:, a]
            self.masks[a][t + 1] = masks[:, a]
            self.active[a][t + 1] = active_masks[:, a]
        self.share[t + 1] = share_obs[:, 0]
        self.rnn_c[t + 1] = rnn_states_critic
        self.values[t] = values
        self.rewards[t] = rewards[:, 0]
        self.cmasks[t + 1] = masks[:, 0]
        self.bad[t + 1] = bad_masks
        self.t = (t + 1) % self.T


def measure(n_envs, steps=40, device=0, warmup=8, env_args=None, devices=None, history_warm_steps=300):
    import gc
    from dc_rl_amd import make_train_env
    from dc_rl_amd import _lib as L
    # (what is already on the heap -- torch, the caller's own objects -- out of the collector's way, as in a fresh runner process:
    # inside bench.py a full collection of that heap is a 35-55 ms pause that lands on whichever loop allocates)
    gc.collect()
    gc.freeze()
    args = {"location": "ny", "days_per_episode": 7, "partial_obs": True, "nonoverlapping_shared_obs_space": True}
    args.update(env_args or {})
    envs = make_train_env("sustaindc", seed=1, n_threads=n_envs, env_args=args, device=device, devices=devices)
    N, k = n_envs, envs.n_agents
    T = 64
    rs = RunnerSide(N, T, envs.observation_space[0].shape[0], envs.share_observation_space[0].shape[0], k)
    rng = np.random.default_rng(0)
    obs, share, avail = envs.reset()
    # (a fresh env's first ~100 steps are slow -- every env rebuilds its young reward windows from a near-empty history ring,
    # 0.3-0.4 ms per launch at 4096 envs: a start-up transient of the first 0.1 % of a history fill, not the loop's rate)
    for _ in range(history_warm_steps):
        envs.step(rng.integers(0, 3, size=(N, k, 1)))
    rnn = np.zeros((N, k, 1, 64), np.float32)
    rnn_c = np.zeros((N, 1, 64), np.float32)
    values = np.zeros((N, 1), np.float32)
    logp = np.zeros((N, k, 1), np.float32)
    t_env = t_all = 0.0
    kept = None
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        actions = rng.integers(0, 3, size=(N, k, 1))                 # collect(): the actors' output shape
        t1 = time.perf_counter()
        obs, share, rew, dones, infos, avail = envs.step(actions)
        t2 = time.perf_counter()
        rs.per_step(infos)
        rs.insert(obs, share, rew, dones, infos, values, actions, logp, rnn, rnn_c)
        t3 = time.perf_counter()
        if it >= warmup:
            t_env += t2 - t1
            t_all += t3 - t0
        kept = infos
    # the runner's own Python: the same walk and inserts over plain materialised dicts (what the reference's runner holds)
    rows = kept.rows()
    cols = list(L.INFO_IDX.items())
    plain = tuple([{**{kk: float(rows[i, j]) for kk, j in cols}, **kept.const[i]} for _ in range(k)] for i in range(N))
    # (the SAME loop with envs.step inside -- its 2.5 MB of fresh outputs evict the runner's working set exactly as in the real
    # loop -- its infos ignored, its time subtracted)
    reps = max(6, steps // 2)
    t_plain = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        actions = rng.integers(0, 3, size=(N, k, 1))
        t1 = time.perf_counter()
        obs, share, rew, dones, _unused, avail = envs.step(actions)
        t2 = time.perf_counter()
        rs.per_step(plain)
        rs.insert(obs, share, rew, dones, plain, values, actions, logp, rnn, rnn_c)
        t_plain += (time.perf_counter() - t0) - (t2 - t1)
    t_plain /= reps
    envs.close()
    gc.unfreeze()
    per = t_all / steps
    return {"envs": N, "steps": steps, "us_per_runner_step": round(per * 1e6, 1), "value": round(N / per, 1), "unit": "env-steps/s",
            "envs_step_us": round(t_env / steps * 1e6, 1), "runner_own_python_us": round(t_plain * 1e6, 1),
            "env_side_share": round(max(0.0, per - t_plain) / per, 4),
            "note": "NumPy actions in / NumPy outputs (PCIe inclusive); logger walk + buffer insert RESTATED from the reference (the "
                    "reference cannot travel to the GPU box); the reference's own SustainDCLogger.per_step + OnPolicyBaseRunner.insert, "
                    "timed beside this restatement on the same inputs in the build container, take 3-11 % longer "
                    "(profiles/r5_harl_reference_loop.txt, tools/harl_reference_loop.py): this rate is an upper bound by that much; "
                    "actor networks not included (they would lower the share)"}


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [48, 512, 4096]
    for n in sizes:
        print(measure(n, steps=40 if n >= 2048 else 100))
