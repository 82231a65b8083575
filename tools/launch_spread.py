import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=24)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 63])
sp = []
for i in range(50):
    o, s, r, d, info = eng.step(pool[i & 63])
    a = info[:, [40, 43]].cpu().numpy()
    pre = info[:, 41].cpu().numpy() / 100.0
    st, en = a[:, 0], a[:, 1]
    if st.max() - st.min() > 500000: continue   # wrapped
    t0 = st.min()
    sp.append((np.percentile(st - t0, [50, 90, 99, 100]), np.percentile(en - t0, [1, 50, 90, 100]), st - t0, pre))
S = np.array([x[0] for x in sp]).mean(0) / 100; E = np.array([x[1] for x in sp]).mean(0) / 100
print("wave start (after gather) relative to first: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(S))
print("wave end: p1 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(E))
st = sp[0][2] / 100
print("start time by env index block of 512:", [round(float(st[i:i+512].mean()), 1) for i in range(0, 4096, 512)])
pre = np.concatenate([x[3] for x in sp])
print("kernel entry -> inputs staged (record load, gather): mean %.2f p50 %.2f p90 %.2f max %.2f us" % (pre.mean(), np.percentile(pre, 50), np.percentile(pre, 90), pre.max()))
