"""Quick steady-state timing of the step (development aid; bench.py is the contract): us per step of sdc_step and of
sdc_rollout (48 steps per launch), 4096 envs (SDC_N), rings full, i.i.d. actions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
N = int(os.environ.get("SDC_N", "4096"))
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=int(os.environ.get("SDC_DBG", "0")))
g = torch.Generator(device="cuda").manual_seed(1234)
POOL = 1024
pool = torch.randint(0, 3, (POOL, N, 3), dtype=torch.int32, device="cuda", generator=g)
eng.reset()
k = 0
for i in range(10300):
    eng.step(pool[k % POOL]); k += 1
res = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3000):
        eng.step(pool[k % POOL]); k += 1
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 3000 * 1e6)
eng.reset()
for i in range(16):
    eng.step(pool[i])
done = 0
torch.cuda.synchronize(); t0 = time.perf_counter()
while done < 1920:
    kk = min(48, eng.steps_to_episode_end())
    o = (16 + done) % (POOL - 48)
    eng.rollout(pool[o:o + kk]); done += kk
torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / done * 1e6
print("N %d  step us %s (best %.2f = %.1f M/s)   rollout us/step %.2f" % (N, ["%.2f" % x for x in res], min(res), N / min(res), tr))
if os.environ.get("SDC_QB_ACTOR", "1") == "1":
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
    for mode in (False, True):
        done = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        while done < 1920:
            kk = min(48, eng.steps_to_episode_end())
            eng.rollout_actor(kk, sample=mode); done += kk
        torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / done * 1e6
        print("closed loop (3 in-kernel actors, %s): %.2f us/step = %.1f M env-steps/s" % ("sampled" if mode else "mode", tr, N / tr))
