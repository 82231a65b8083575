"""Where the closed-loop kernel's time goes (development aid; needs a library built with -DSDC_ACTOR_CLOCK, which makes
sdc_rollout_actor_kernel sum shader-clock cycles per phase over the launch's K steps into info slots 38..40 of the last step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
N = int(os.environ.get("SDC_N", "4096"))
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234)
g = torch.Generator(device="cuda").manual_seed(1234)
pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, device="cuda", generator=g)
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
K = 48
for rep in range(6):
    out = eng.rollout_actor(K)
info = out["info"] if isinstance(out, dict) else out[4]
ck = info[K - 1].reshape(N, -1)[0::2, 38:41].double().cpu().numpy() / K          # per pair wavefront, cycles per step
for i, nm in enumerate(["networks", "pick", "env step"]):
    c = ck[:, i]
    print("  %-10s mean %8.0f  p50 %8.0f  p99 %8.0f cycles/step" % (nm, c.mean(), np.median(c), np.percentile(c, 99)))
print("  total %.0f cycles/step" % ck.sum(axis=1).mean())
