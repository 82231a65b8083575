"""-DSDC_WIDE_STAMPS build: what the LAST-ENDING workgroup of every launch of the lane-per-env kernel did differently -- the reward
wavefront's segments before barrier 2 (headers in LDS -> arrivals served -> evictions applied -> oldest task found), behind it, and when
the workgroup entered.  usage: python tools/dev/wide_tail.py N"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
N = int(sys.argv[1])
e, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=2048)
g = torch.Generator(device="cpu").manual_seed(1234)
acts = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
e.reset()
for t in range(10300): e.step(acts[t % 64])
acc = []
for t in range(200):
    e.step(acts[t % 64]); acc.append(e.info[::64, :24].cpu().numpy().astype(np.int64).copy())
a = np.stack(acc)
top = a[..., 16]
first = top.min(1, keepdims=True)
rel = lambda x: ((x - first) % (1 << 24)) / 100.0
T = {k: rel(a[..., k]) for k in range(24)}
end = T[15]
last = end.argmax(1)
L = np.arange(a.shape[0])
segs = [("first instruction (after the launch's first)", None, 16), ("-> D entry (arguments read)", 16, 0), ("-> R headers in LDS, queue reads issued", 0, 8),
        ("-> R header in registers, 16 window keys requested", 8, 18), ("-> R arrivals served", 18, 17), ("-> R evictions applied", 17, 9), ("-> R oldest task found (at barrier 2)", 9, 19),
        ("   D energy handed over (at barrier 2), after D entry", 0, 4), ("barrier 2 -> R insertions applied", 10, 11), ("-> R moments", 11, 12),
        ("-> R requests filed, committed", 12, 13), ("-> R header out, fallbacks", 13, 14), ("-> R info out", 14, 15)]
print(f"N={N}, 200 launches x {a.shape[1]} workgroups: segment (us): all workgroups mean / p99 | the launch's LAST-ENDING workgroup mean / p90")
for nm, x, y in segs:
    d = T[y] if x is None else T[y] - T[x]
    dl = d[L, last]
    print(f"  {nm:58s} {d.mean():6.2f} {np.percentile(d, 99):6.2f} | {dl.mean():6.2f} {np.percentile(dl, 90):6.2f}")
print("  end after the launch's first instruction: all mean %.2f p99 %.2f | last-ending mean %.2f p90 %.2f" % (end.mean(), np.percentile(end, 99), end[L, last].mean(), np.percentile(end[L, last], 90)))
# without the k slowest workgroups of every launch
s = np.sort(end, axis=1)
print("  launch end without its k slowest workgroups (mean over launches): " + " ".join("k=%d %.2f" % (k, s[:, -1 - k].mean()) for k in (0, 1, 2, 4, 8, 16, 32, 64)))
