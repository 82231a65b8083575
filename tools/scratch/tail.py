import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from dc_rl_amd import _lib as L
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=8)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 255])
D, R = [], []
for i in range(200):
    o, s, r, d, info = eng.step(pool[i & 255])
    a = info[::2, 41:44].cpu().numpy() / 100.0
    D.append(a[:, 0]); R.append(a[:, 1])
D = np.stack(D); R = np.stack(R)   # [launch, wave]
print("dyn mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % ((D.mean(),) + tuple(np.percentile(D, [50, 90, 99, 100]))))
print("rew mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % ((R.mean(),) + tuple(np.percentile(R, [50, 90, 99, 100]))))
print("corr(dyn, rew) %.3f" % np.corrcoef(D.ravel(), R.ravel())[0, 1])
# per-wave-index persistent slowness (same wave slow in every launch -> placement), vs per-launch random
print("std of per-wave mean dyn %.3f, of per-launch-residual %.3f" % (D.mean(0).std(), (D - D.mean(0)).std()))
wm = D.mean(0) + R.mean(0)
order = np.argsort(-wm)
print("slowest wave indices (mean total):", [(int(k), round(float(wm[k]), 2)) for k in order[:16]])
print("fastest:", [(int(k), round(float(wm[k]), 2)) for k in order[-8:]])
# by block position: pairs -> block = vb; hardware block index bi: vb = (bi%8)*(nb/8)+bi/8
nb = N // 2 // 4
vb = np.arange(N // 2) // 4
bi = (vb % (nb // 8)) * 8 + vb // (nb // 8)
for lo in range(0, nb, 128):
    m = (bi >= lo) & (bi < lo + 128)
    print("hw blocks %3d-%3d: dyn %.2f rew %.2f" % (lo, lo + 127, D[:, m].mean(), R[:, m].mean()))
for x in range(8):
    m = (bi % 8) == x
    print("xcd %d: dyn %.2f rew %.2f tot p99 %.2f" % (x, D[:, m].mean(), R[:, m].mean(), np.percentile(D[:, m] + R[:, m], 99)))
