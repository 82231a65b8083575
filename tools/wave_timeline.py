"""Diagnostic: the life of every wavefront of a step launch on one time axis (in-kernel clock stamps, debug_flags 8 + 16):
entry, inputs staged, end; which wavefronts end last, and what the launch span would be without the slowest ones."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=24)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 255])
acc = []
for i in range(200):
    o, s, r, d, info = eng.step(pool[i & 255])
    a = info[::2, 39:44].cpu().numpy().copy()
    st, pre, en = a[:, 1], a[:, 2], a[:, 4]
    if st.max() - st.min() > 500000 or en.max() < st.min(): continue
    t0 = (st - pre).min()
    acc.append(np.stack([st - pre - t0, st - t0, en - t0, info[::2, 39].cpu().numpy(), info[1::2, 39].cpu().numpy()], 1) )
A = np.stack(acc)   # [launch, wave, 5]  (10 ns units)
A[:, :, :3] /= 100.0
L_, W, _ = A.shape
entry, staged, end = A[:, :, 0], A[:, :, 1], A[:, :, 2]
dur = end - staged
print("launches", L_, "waves", W)
print("entry  p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(entry, [50, 90, 100], axis=1).mean(1)))
print("staged p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(staged, [50, 90, 100], axis=1).mean(1)))
print("end    p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(end, [50, 90, 99, 100], axis=1).mean(1)))
print("dur    mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % ((dur.mean(),) + tuple(np.percentile(dur, [50, 90, 99, 100], axis=1).mean(1))))
# per launch: the last-ending wave: its entry, its duration, its index
li = end.argmax(1)
r = np.arange(L_)
print("last-ending wave: entry %.2f staged %.2f dur %.2f end %.2f ; index/256 hist" % (entry[r, li].mean(), staged[r, li].mean(), dur[r, li].mean(), end[r, li].mean()), np.bincount(li // 256, minlength=8))
print("  reward path of that wave (max of pair):", np.bincount(np.maximum(A[r, li, 3], A[r, li, 4]).astype(int), minlength=4))
# what would the span be without the top-k longest-duration waves
for k in (1, 5, 20, 100):
    e2 = end.copy()
    idx = np.argsort(-dur, axis=1)[:, :k]
    for j in range(L_): e2[j, idx[j]] = 0
    print("span without the %d longest waves: %.2f" % (k, e2.max(1).mean()))
# by wave index: mean entry / dur by block group
for lo in range(0, W, 256):
    print("waves %4d-%4d: entry %.2f staged %.2f dur %.2f end %.2f" % (lo, lo + 255, entry[:, lo:lo+256].mean(), staged[:, lo:lo+256].mean(), dur[:, lo:lo+256].mean(), end[:, lo:lo+256].mean()))
# duration vs reward path
pth = np.maximum(A[:, :, 3], A[:, :, 4]).astype(int)
for p_ in range(4):
    m = pth == p_
    if m.sum(): print("path %d: share %.4f dur %.2f" % (p_, m.mean(), dur[m].mean()))
# correlation between dur and entry (late waves slower?)
print("corr(entry, dur) %.3f" % np.corrcoef(entry.ravel(), dur.ravel())[0, 1])
for q in (0, 25, 50, 75, 90, 99):
    lo_, hi_ = np.percentile(entry, q), np.percentile(entry, min(q + 10, 100))
    m = (entry >= lo_) & (entry <= hi_)
    print("entry in [%.2f, %.2f]: dur mean %.2f p99 %.2f" % (lo_, hi_, dur[m].mean(), np.percentile(dur[m], 99)))
