import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=8 + 16 + 32)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 255])
A = []
for i in range(100):
    o, s, r, d, info = eng.step(pool[i & 255])
    A.append(info[::2, 40:42].cpu().numpy() / 100.0)
A = np.concatenate(A)
print("entry -> record arrived: mean %.2f p90 %.2f max %.2f ; entry -> staged: mean %.2f p90 %.2f" % (A[:, 0].mean(), np.percentile(A[:, 0], 90), A[:, 0].max(), A[:, 1].mean(), np.percentile(A[:, 1], 90)))
