"""us per sdc_step launch (no episode boundary inside the timed window) by batch size and lane mapping (development aid):
debug_flags 512 = two envs per wavefront, 1024 = four."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for N in [int(x) for x in os.environ.get("SDC_NS", "4096,8192,12288,16384").split(",")]:
    for flags in (512, 1024):
        eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=flags)
        g = torch.Generator(device="cuda").manual_seed(1234)
        pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, device="cuda", generator=g)
        eng.reset()
        for i in range(120):
            eng.step(pool[i % 256])
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(150):
                eng.step(pool[(i + 7 * rep) % 256])
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 150 * 1e6)
        done = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        while done < 384:
            kk = min(48, eng.steps_to_episode_end())
            eng.rollout(pool[:kk]); done += kk
        torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / done * 1e6
        import numpy as np
        rngw = np.random.default_rng(7)
        for a_ in range(3):
            eng.set_actor(a_, {"ln0_gamma": 1 + 0.1 * rngw.standard_normal(26), "ln0_beta": 0.1 * rngw.standard_normal(26),
                               "w1": rngw.standard_normal((64, 26)) * 0.3, "b1": 0.1 * rngw.standard_normal(64),
                               "ln1_gamma": 1 + 0.1 * rngw.standard_normal(64), "ln1_beta": 0.1 * rngw.standard_normal(64),
                               "w2": rngw.standard_normal((64, 64)) * 0.2, "b2": 0.1 * rngw.standard_normal(64),
                               "ln2_gamma": 1 + 0.1 * rngw.standard_normal(64), "ln2_beta": 0.1 * rngw.standard_normal(64),
                               "w3": rngw.standard_normal((3, 64)) * 0.2, "b3": np.zeros(3), "activation": "tanh"})
        eng.reset()
        for i in range(16):
            eng.step(pool[i])
        done = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        while done < 480:
            kk = min(48, eng.steps_to_episode_end())
            eng.rollout_actor(kk, sample=True); done += kk
        torch.cuda.synchronize(); tc = (time.perf_counter() - t0) / done * 1e6
        print("N %6d  %s: %.2f us per step = %.1f M env-steps/s;  sdc_rollout (48 per launch) %.2f us = %.1f M;  closed loop (sampled) %.2f us = %.1f M" % (
            N, "two envs per wavefront " if flags == 512 else "four envs per wavefront", best, N / best, tr, N / tr, tc, N / tc))
        eng.close()
