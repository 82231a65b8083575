"""Diagnostic: histogram of the reward path codes (info[:, 39]) over steady-state steps (debug_flags bit 1 = miss reasons)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
N = int(os.environ.get("SDC_N", "4096"))
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=int(os.environ.get("SDC_DBG","2")))
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
mode = os.environ.get("SDC_MODE", "rand")
if mode == "ls1": pool[:, :, 0] = 1
if mode == "bat2": pool[:, :, 2] = 2
if mode == "idle": pool[:, :, 0] = 1; pool[:, :, 1] = 1; pool[:, :, 2] = 2
eng.reset()
for i in range(10000 + 300):
    eng.step(pool[i & 63])
h = np.zeros(1300, np.int64)
missed = {}
for i in range(int(os.environ.get("SDC_STEPS", "400"))):
    o, s, r, d, info = eng.step(pool[i & 63])
    p = info[:, 39].cpu().numpy().astype(int)
    h += np.bincount(p, minlength=1300)[:1300]
    for e in np.nonzero((p >= 3) & (p < 1000))[0]:
        missed.setdefault(int(e), []).append((i, int(p[e])))
print("path codes per step:", {k: round(v / int(os.environ.get("SDC_STEPS", "400")), 3) for k, v in enumerate(h) if v})
print("envs with misses:", len(missed), "examples:", list(missed.items())[:3])
hist = eng.get_state("hist")
for e in list(missed)[:3]:
    v = np.sort(hist[e][~np.isnan(hist[e])])
    q1, q3 = np.percentile(v.astype(np.float64), [25, 75]); ub = q3 + 1.5 * (q3 - q1)
    top = v[v > ub - 3 * (q3 - q1) * 0.05]
    u, c = np.unique(v, return_counts=True)
    print("env", e, "n", len(v), "ub", ub, "n>ub", (v > ub).sum(), "max dup count", c.max(), "dups near ub:", [(float(a), int(b)) for a, b in zip(u, c) if abs(a - ub) < 2.0 and b > 1][:10])
    near = v[np.abs(v - ub) < 1.0]
    print("   keys within 1.0 of ub:", len(near), near[:12])
