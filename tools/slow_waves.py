"""Diagnostic: which wavefronts of a launch live longest?  Per-wavefront lifetime (in-kernel clock stamps, debug_flags
bit 3) against what its two envs did in that step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from dc_rl_amd import _lib as L
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=8)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 255])
rows, acts = [], []
for i in range(80):
    a = pool[i & 255]
    o, s, r, d, info = eng.step(a)
    rows.append(info.cpu().numpy().copy()); acts.append(a.cpu().numpy().copy())
I = np.concatenate(rows); A = np.concatenate(acts)
tot = I[0::2, 43] / 100.0; dyn = I[0::2, 41] / 100.0; rw = I[0::2, 42] / 100.0
def pair_any(x): return x[0::2] | x[1::2]
ix = L.INFO_IDX
feat = {
  "ls defer (a=0)": pair_any(A[:, 0] == 0), "ls process (a=2)": pair_any(A[:, 0] == 2),
  "tasks processed > 0": pair_any(I[:, ix["ls_tasks_processed"]] > 0), "overdue > 0": pair_any(I[:, ix["ls_overdue_penalty"]] > 0),
  "queue non-empty": pair_any(I[:, ix["ls_tasks_in_queue"]] > 0),
  "bat idle both": (A[0::2, 2] == 2) & (A[1::2, 2] == 2), "bat differs": A[0::2, 2] != A[1::2, 2],
  "reward path != 0": pair_any(I[:, 39] != 0), "path 2 (takeover)": pair_any(I[:, 39] == 2),
  "request filed": pair_any(I[:, 39] == 4),
}
print("all waves: total %.2f dyn %.2f rew %.2f   p99 %.2f  max %.2f" % (tot.mean(), dyn.mean(), rw.mean(), np.percentile(tot, 99), tot.max()))
for k, m in feat.items():
    if m.sum() == 0: continue
    print("%-22s share %.3f  total %.2f (else %.2f)  dyn %.2f (else %.2f)  rew %.2f (else %.2f)" % (
        k, m.mean(), tot[m].mean(), tot[~m].mean(), dyn[m].mean(), dyn[~m].mean(), rw[m].mean(), rw[~m].mean()))
slow = tot >= np.percentile(tot, 99)
print("slowest 1%%: dyn %.2f rew %.2f;" % (dyn[slow].mean(), rw[slow].mean()), {k: round(float(m[slow].mean()), 2) for k, m in feat.items()})
w = np.arange(len(tot)) % (N // 2)
print("slowest 1%% by wave index / 256:", np.bincount(w[slow] // 256, minlength=8))
