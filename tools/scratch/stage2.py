import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=8 + 16 + 32)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(1000): eng.step(pool[i & 255])
A = []
for i in range(50):
    o, s, r, d, info = eng.step(pool[i & 255])
    A.append(info[::2, 39:44].cpu().numpy() / 100.0)
A = np.concatenate(A)
print("from entry: record %.2f | fields read %.2f | before fast block %.2f | after fast block %.2f | staged %.2f" % (A[:, 1].mean(), A[:, 0].mean(), A[:, 3].mean(), A[:, 4].mean(), A[:, 2].mean()))
