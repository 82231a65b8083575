"""Diagnostic (measurement build -DSDC_FAST_DEBUG=1): which wavefronts of a step launch are the slow ones -- duration
by rare path taken (info[reserved] bits), by a_ls of the pair, by where they ran (SIMD shared with how many others)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=8 | 16 | 256)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (1024, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
k = 0
for i in range(10300):
    eng.step(pool[k % 1024]); k += 1
rows = []
for i in range(300):
    a = pool[k % 1024]; k += 1
    o, s, r, d, info = eng.step(a)
    inf = info.cpu().numpy().copy()
    st, pre, en = inf[::2, 40 + 0], inf[::2, 41], inf[::2, 43]
    rows.append((inf[::2, 39].astype(int), inf[1::2, 39].astype(int), inf[::2, 40].astype(int), inf[::2, 41], inf[::2, 43], inf[::2, 42],
                 a[::2, 0].cpu().numpy(), a[1::2, 0].cpu().numpy()))
P0 = np.stack([r[0] for r in rows]); P1 = np.stack([r[1] for r in rows]); HW = np.stack([r[2] for r in rows])
PRE = np.stack([r[3] for r in rows]) / 100.0; END = np.stack([r[4] for r in rows]); REW = np.stack([r[5] for r in rows]) / 100.0
A0 = np.stack([r[6] for r in rows]); A1 = np.stack([r[7] for r in rows])
# (flag 16 without 32: inf[40] would be dbg_a0 & 0xFFFFF; with 256 it is the HW id; inf[43] = end & 0xFFFFF, inf[41] = entry -> staged)
bits = (P0 >> 3) | (P1 >> 3)
path = np.maximum(P0 & 7, P1 & 7)
print("waves x launches", P0.shape[1], P0.shape[0])
# duration proxy: reward part + (no absolute start here) -> use inf[42] (reward) and the END order inside the launch
end = (END - END.min(1, keepdims=True)) / 100.0
end = np.where(end > 1000, np.nan, end)
print("end rel. to first end: p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.nanpercentile(end, [50, 90, 99, 100], axis=1).mean(1)))
def show(name, m):
    if m.sum() == 0: return
    print("%-34s share %.4f  end-rel mean %.2f  reward part %.2f us" % (name, m.mean(), np.nanmean(end[m]), REW[m].mean()))
show("all", np.ones_like(bits, bool))
show("plain (no rare path, no a_ls=2)", (bits == 0) & (A0 != 2) & (A1 != 2) & (path == 0))
show("a_ls = 2 in one env", ((A0 == 2) ^ (A1 == 2)))
show("a_ls = 2 in both envs", ((A0 == 2) & (A1 == 2)))
for b, nm in ((1, "oldest-task table search"), (2, "key inside a window"), (4, "bound crossed keys"), (8, "deferred window arrived"), (16, "request filed")):
    show(nm, (bits & b) != 0)
# co-residency: how many of the launch's pair wavefronts share the SIMD (xcc, se, sh, cu, simd)
simd = HW >> 4          # drop the wave slot id
last = np.nanargmax(end, axis=1)
cnt = np.zeros_like(bits)
for j in range(P0.shape[0]):
    u, inv, c = np.unique(simd[j], return_inverse=True, return_counts=True)
    cnt[j] = c[inv]
for c in range(1, 6):
    show("pair wavefronts on its SIMD = %d" % c, cnt == c)
r = np.arange(P0.shape[0])
print("last-ending wavefront: bits hist", np.bincount(bits[r, last], minlength=32)[:32], " a_ls2 any %.2f" % ((A0[r, last] == 2) | (A1[r, last] == 2)).mean(),
      " on-SIMD count hist", np.bincount(cnt[r, last], minlength=6))
print("distinct SIMDs used per launch: %.1f" % np.mean([len(np.unique(simd[j])) for j in range(P0.shape[0])]))
