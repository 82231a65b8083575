import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=0)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 63])
rows = []
for i in range(30):
    o, s, r, d, info = eng.step(pool[i & 63])
    p = info[:, 39].cpu().numpy().astype(int)
    hd = eng.get_state("header")
    for e in np.nonzero(p == 1)[0]:
        rows.append(hd[e, 18:28].astype(np.int64))
rows = np.array(rows)
print("refills", len(rows), "per step", len(rows) / 30)
print("cols: rounds m sweep_ticks rank_ticks total_ticks filled kept dir")
print(rows[:12, :8])
print("rounds hist", np.bincount(rows[:, 0]))
one = rows[rows[:, 0] == 1]
print("single-round: sweep us %.2f  compact+rank+place us %.2f  total us %.2f  m mean %.1f  added mean %.1f" % (one[:, 2].mean() / 100, one[:, 3].mean() / 100, one[:, 4].mean() / 100, one[:, 1].mean(), (one[:, 5] - one[:, 6]).mean()))
print("all: total us %.2f" % (rows[:, 4].mean() / 100))
