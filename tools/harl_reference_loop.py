#!/usr/bin/env python3
"""The reference's OWN runner code timed over this package's `infos` (CPU only; build container only -- it imports
/root/reference): `SustainDCLogger.per_step` (harl/envs/sustaindc/sustaindc_logger.py:80-101, with `BaseLogger.per_step`,
harl/common/base_logger.py:43-64) and `OnPolicyBaseRunner.insert` (harl/runners/on_policy_base_runner.py:386-503) writing into
the reference's own `OnPolicyActorBuffer` / `OnPolicyCriticBufferEP`, fed one step's outputs of an N-env batch:
  * `infos` = the product's C-typed `LazyInfos` over a float32 [N, 44] host block (what `SustainDCVecEnv.step` returns in NumPy
    mode), and -- for comparison -- plain materialised dicts (what the reference's own vector env returns);
  * beside it the RESTATEMENT of the same two functions that `tools/harl_loop_rate.py` (bench.py `secondary.harl_unchanged_loop`)
    times on the GPU box, where the reference cannot travel, on the same inputs on the same cores.
Third-party packages this image lacks (absl, setproctitle, tensorboardX, gymnasium, ...) are stood in for by
tests/aux/harl_stubs and tests/golden/_shims; the runner object is created without its constructor (which builds envs and
networks) and given exactly the attributes `insert` reads.
usage: python tools/harl_reference_loop.py [N ...]      (default 4096 512 48)"""
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SDC_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(ROOT, "tests", "aux", "harl_stubs"), os.path.join(ROOT, "tests", "golden", "_shims"), REF, ROOT]
sys.path.insert(0, os.path.join(ROOT, "tools"))


def info_block(N, rng):
    """A float32 [N, 44] info block with the magnitudes of a real step."""
    from dc_rl_amd import _lib as L
    b = np.zeros((N, L.INFO_DIM), np.float32)
    put = lambda k, v: b.__setitem__((slice(None), L.INFO_IDX[k]), v)
    put("bat_total_energy_with_battery_KWh", 331 + 70 * rng.standard_normal(N))
    put("bat_CO2_footprint", 9e4 + 1e4 * rng.standard_normal(N))
    put("dc_water_usage", 300 + 30 * rng.standard_normal(N))
    put("ls_tasks_in_queue", rng.integers(0, 400, N))
    put("ls_tasks_dropped", rng.integers(0, 3, N))
    put("dc_ITE_total_power_kW", 1000 + 100 * rng.standard_normal(N))
    put("dc_CT_total_power_kW", 60 + 5 * rng.standard_normal(N))
    put("dc_Compressor_total_power_kW", 250 + 20 * rng.standard_normal(N))
    put("dc_HVAC_total_power_kW", 310 + 20 * rng.standard_normal(N))
    return b


def measure(N, steps=30, warmup=5):
    import harl_loop_rate as restated
    from harl.common.buffers.on_policy_actor_buffer import OnPolicyActorBuffer
    from harl.common.buffers.on_policy_critic_buffer_ep import OnPolicyCriticBufferEP
    from harl.envs.sustaindc.sustaindc_logger import SustainDCLogger
    from harl.runners.on_policy_base_runner import OnPolicyBaseRunner
    from dc_rl_amd import _lib as L
    from dc_rl_amd.spaces import Box, Discrete
    from dc_rl_amd.vec_env import LazyInfos

    k, T, H = 3, 64, 64
    rng = np.random.default_rng(0)
    args = {"episode_length": T, "n_rollout_threads": N, "hidden_sizes": [H, H], "recurrent_n": 1, "gamma": 0.99, "gae_lambda": 0.95,
            "use_gae": True, "use_proper_time_limits": True}
    obs_space = Box(low=-2.0, high=2.0, shape=(26,), dtype=np.float32)
    share_space = Box(low=-2.0, high=2.0, shape=(29,), dtype=np.float32)
    runner = object.__new__(OnPolicyBaseRunner)            # the class unchanged; no constructor (it builds envs and networks)
    runner.num_agents, runner.recurrent_n, runner.rnn_hidden_size, runner.state_type = k, 1, H, "EP"
    runner.algo_args = {"train": {"n_rollout_threads": N}}
    runner.actor_buffer = [OnPolicyActorBuffer(args, obs_space, Discrete(3)) for _ in range(k)]
    runner.critic_buffer = OnPolicyCriticBufferEP(args, share_space)
    logger = object.__new__(SustainDCLogger)
    logger.algo_args = {"train": {"n_rollout_threads": N}}
    logger.train_episode_rewards = np.zeros(N)
    logger.done_episodes_rewards = []
    SustainDCLogger.episode_init(logger, 0) if False else None
    logger.metrics = {m: 0 for m in ("net_energy_sum", "ite_power_sum", "ct_power_sum", "chiller_power_sum", "hvac_power_sum",
                                     "CO2_footprint_sum", "water_usage", "step_count", "load_left", "ls_tasks_in_queue",
                                     "ls_tasks_dropped", "PUE")}
    logger.metrics.update(instantaneous_net_energy=[], hvac_power_on_used=[])
    rs = restated.RunnerSide(N, T, 26, 29, k)

    const = [{"ls_queue_max_len": 1000, "ls_norm_load_left": 0, "ls_unasigned_day_load_left": 0, "ls_penalty_flag": 0,
              "ls_enforced": 0, "dc_power_lb_kW": 300.0, "dc_power_ub_kW": 4800.0, "dc_CW_pump_power_kW": 1.0,
              "dc_CT_pump_power_kW": 1.0, "bat_max_bat_cap": 4.8, "bat_dcload_min": 75.0, "bat_dcload_max": 1200.0}] * N
    obs = rng.standard_normal((N, k, 26)).astype(np.float32)
    share = np.broadcast_to(rng.standard_normal((N, 1, 29)).astype(np.float32), (N, k, 29))
    rew = rng.standard_normal((N, k, 1)).astype(np.float32)
    dones = np.zeros((N, k), bool)
    avail = np.ones((N, k, 3), np.float32)
    values = np.zeros((N, 1), np.float32)
    logp = np.zeros((N, k, 1), np.float32)
    rnn = np.zeros((N, k, 1, H), np.float32)
    rnn_c = np.zeros((N, 1, H), np.float32)
    done_h = np.zeros(N, bool)
    cols = list(L.INFO_IDX.items())

    def run(kind, which):
        tot = 0.0
        for it in range(warmup + steps):
            block = info_block(N, rng)
            actions = rng.integers(0, 3, size=(N, k, 1))
            if kind == "lazy":
                infos = LazyInfos(block, actions.reshape(N, k), done_h, const, {}, None, 0, k)
            else:
                infos = tuple([{**{kk: float(block[i, j]) for kk, j in cols}, **const[i]} for _ in range(k)] for i in range(N))
            data = (obs, share, rew, dones, infos, avail, values, actions, logp, rnn, rnn_c)
            t0 = time.perf_counter()
            if which == "reference":
                logger.per_step(data)
                runner.insert(data)
            else:
                rs.per_step(infos)
                rs.insert(obs, share, rew, dones, infos, values, actions, logp, rnn, rnn_c)
            dt = time.perf_counter() - t0
            if which == "reference":              # (the reference's buffers advance a step counter modulo the episode length themselves)
                pass
            if it >= warmup:
                tot += dt
        return tot / steps

    gc.collect()
    gc.freeze()
    out = {"envs": N}
    for kind in ("lazy", "dicts"):
        for which in ("reference", "restated"):
            out[f"{which}_{kind}_us"] = round(run(kind, which) * 1e6, 1)
    gc.unfreeze()
    out["us_per_env_reference_lazy"] = round(out["reference_lazy_us"] / N, 3)
    out["us_per_env_restated_lazy"] = round(out["restated_lazy_us"] / N, 3)
    out["restated_over_reference"] = round(out["restated_lazy_us"] / out["reference_lazy_us"], 3)
    return out


if __name__ == "__main__":
    print("cores:", os.cpu_count(), "| per runner step: the reference's own logger.per_step + runner.insert vs tools/harl_loop_rate.py's restatement,"
          " over LazyInfos (the product's infos) and over plain dicts")
    for n in [int(x) for x in sys.argv[1:]] or [4096, 512, 48]:
        print(measure(n, steps=30 if n >= 2048 else 100))
