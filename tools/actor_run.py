"""A few closed-loop launches (sdc_rollout_actor, 48 steps each, 4096 envs) for a profiler to look at (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
N = int(os.environ.get("SDC_N", "4096"))
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234)
g = torch.Generator(device="cuda").manual_seed(1234)
pool = torch.randint(0, 3, (16, N, 3), dtype=torch.int32, device="cuda", generator=g)
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
for rep in range(8):
    eng.rollout_actor(48)
torch.cuda.synchronize()
