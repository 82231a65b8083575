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
print("cols: tries m extra need bw hi hi2 dir s span")
print(rows[:40])
print("tries hist", np.bincount(rows[:, 0]))
print("dir hist", np.bincount(rows[:, 7]), "sweep us", rows[:, 5].mean() / 100, "rank us", rows[:, 6].mean() / 100, "assemble us", rows[:, 8].mean() / 100, "m mean", rows[:, 1].mean())
