"""Diagnostic (debug_flags bit 3): per-wavefront phase durations of the step kernel, from in-kernel wall-clock stamps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=int(os.environ.get("SDC_DBG", "8")))
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 63])
rows = []
for i in range(100):
    o, s, r, d, info = eng.step(pool[i & 63])
    rows.append(info[:, 39:44].cpu().numpy().copy())
a = np.concatenate(rows)
for name, m in (("all", a[:, 0] >= 0), ("no-refill", a[:, 0] == 0), ("refill", a[:, 0] == 1)):
    x = a[m][:, 1:] / 100.0
    print("%-9s n/step %6.1f  ahead %5.2f  dynamics %5.2f  reward %5.2f  total %5.2f us (mean)   total p99 %5.2f max %5.2f" % (
        name, m.sum() / 100, x[:, 0].mean(), x[:, 1].mean(), x[:, 2].mean(), x[:, 3].mean(), np.percentile(x[:, 3], 99), x[:, 3].max()))
tot = a[:, 4] / 100.0
thr = np.percentile(tot[tot < 100], 99)
m = (tot >= thr) & (tot < 100)
x = a[m][:, 1:] / 100.0
print("slowest 1%% (n/step %.1f): ahead %5.2f  dynamics %5.2f  reward %5.2f  total %5.2f | refill share %.2f" % (
    m.sum() / 100, x[:, 0].mean(), x[:, 1].mean(), x[:, 2].mean(), x[:, 3].mean(), (a[m][:, 0] == 1).mean()))
env_of = np.tile(np.arange(N), 100)[m]
print("slowest 1%%: env index histogram by 512-block:", np.bincount(env_of // 512, minlength=8))
