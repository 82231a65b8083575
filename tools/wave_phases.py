import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
N = int(os.environ.get("SDC_N", "4096"))
for flags in (8, 24):
    eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=flags)
    g = torch.Generator(device="cpu").manual_seed(1234)
    pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
    eng.reset()
    for i in range(10300): eng.step(pool[i & 255])
    rows = []
    for i in range(60):
        o, s, r, d, info = eng.step(pool[i & 255])
        rows.append(info[::2, 39:44].cpu().numpy().copy())
    a = np.concatenate(rows)
    if flags == 8:
        x = a[:, 2:] / 100.0
        print("per wave (us): dynamics %.2f  reward(2 envs) %.2f  total-after-gather %.2f | p99 total %.2f max %.2f" % (
            x[:, 0].mean(), x[:, 1].mean(), x[:, 2].mean(), np.percentile(x[:, 2], 99), x[:, 2].max()))
    else:
        st, pre, en = a[:, 1], a[:, 2] / 100.0, a[:, 4]
        per = len(a) // 60
        S, E = [], []
        for k in range(60):
            s_, e_ = st[k*per:(k+1)*per], en[k*per:(k+1)*per]
            if s_.max() - s_.min() > 500000 or e_.max() < s_.min(): continue
            t0 = s_.min(); S.append(np.percentile(s_ - t0, [50, 90, 99, 100]) / 100); E.append(np.percentile(e_ - t0, [1, 50, 90, 100]) / 100)
        print("wave start (inputs staged) rel. to first: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.mean(S, 0)))
        print("wave end: p1 %.1f p50 %.1f p90 %.1f max %.1f us" % tuple(np.mean(E, 0)))
        print("entry -> inputs staged: mean %.2f p90 %.2f us" % (pre.mean(), np.percentile(pre, 90)))
    eng.close()
