"""Where a NumPy-mode SustainDCVecEnv.step spends its time when the caller is slow (a Python runner: the GPU idles ~10 ms
between steps) against a tight loop.  usage: python tools/np_step_profile.py [N] [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dc_rl_amd import make_train_env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
args = {"location": "ny", "days_per_episode": 7, "partial_obs": True, "nonoverlapping_shared_obs_space": True}
envs = make_train_env("sustaindc", seed=1, n_threads=N, env_args=args)
envs.reset()
rng = np.random.default_rng(0)
for _ in range(300):
    envs.step(rng.integers(0, 3, size=(N, 3, 1)))      # past the young-history transient
for mode in ("tight", "idle"):
    acc = np.zeros(6)
    n = 60
    for it in range(n + 10):
        a = rng.integers(0, 3, size=(N, 3, 1))
        if mode == "idle":
            time.sleep(idle * 1e-3)
        t0 = time.perf_counter()
        envs.step_async(a)
        t1 = time.perf_counter()
        e = envs.engine
        act = envs._actions; envs._actions = None
        e.step(act)
        t2 = time.perf_counter()
        hb = envs._host_buffers()
        hb["flat"].copy_(e.out_flat, non_blocking=True)
        t3 = time.perf_counter()
        torch.cuda.current_stream(e.device).synchronize()
        t4 = time.perf_counter()
        from dc_rl_amd.vec_env import LazyInfos
        done_h = hb["done"].numpy().astype(bool)
        infos = LazyInfos(hb["info"], act, done_h, envs._const, {}, None, 1, 3)
        t5 = time.perf_counter()
        x = infos[0][0].get("bat_SOC")
        t6 = time.perf_counter()
        if it >= 10:
            acc += np.diff([t0, t1, t2, t3, t4, t5, t6])
    print(mode, "us: step_async %.1f  launch %.1f  copy-enqueue %.1f  sync %.1f  infos-init %.1f  first-access %.1f  total %.1f" %
          (*(acc / n * 1e6), acc.sum() / n * 1e6))
envs.close()
