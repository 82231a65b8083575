import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
N = 4096
for flags in (8, 8 + 64, 8 + 16 + 32, 8 + 16 + 32 + 64):
    eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=flags)
    g = torch.Generator(device="cpu").manual_seed(1234)
    pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
    eng.reset()
    for i in range(300): eng.step(pool[i & 255])
    A = []
    for i in range(30):
        o, s, r, d, info = eng.step(pool[i & 255])
        A.append(info[::2, 40:44].cpu().numpy() / 100.0)
    A = np.concatenate(A)
    if flags & 16:
        print("flags %3d: entry->record %.2f  entry->staged %.2f" % (flags, A[:, 0].mean(), A[:, 1].mean()))
    else:
        print("flags %3d: dyn %.2f rew %.2f total %.2f" % (flags, A[:, 1].mean(), A[:, 2].mean(), A[:, 3].mean()))
    eng.close()
