"""Development aid: check the quartile-tracker windows against the sorted ring on the host."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import parity_util as P
from dc_rl_amd import _lib as L

def f32_key(f):
    b = np.asarray(f, np.float32).view(np.uint32).astype(np.uint64)
    neg = (b >> 31) & 1
    return np.where(neg == 1, (~b) & 0xFFFFFFFF, b | 0x80000000).astype(np.uint64)

N = 8
rig = P.ParityRig(N, episode_steps=672, seed=5, with_oracle=False, debug_flags=1) if "debug_flags" in P.ParityRig.__init__.__code__.co_varnames else P.ParityRig(N, episode_steps=672, seed=5, with_oracle=False)
eng = rig.eng
rng = np.random.default_rng(3)
hist = np.full((N, eng.hist_stride), np.nan, np.float32)
n0 = int(os.environ.get("N0", "10000"))
vals = (331 + 70 * rng.standard_normal((N, n0))).astype(np.float32)
hist[:, :n0] = vals
eng.set_state("hist", hist)
eng.set_state("hist_len", np.full(N, n0, np.int32))
eng.set_state("hist_pos", np.zeros(N, np.int32))
rig.reset_all()
steps = int(os.environ.get("STEPS", "3"))
for t in range(steps):
    a = torch.randint(0, 3, (N, 3), dtype=torch.int32, device=eng.device)
    obs, share, rew, done, info = eng.step(a)
    inf = info.cpu().numpy()
    hd = eng.get_state("header"); qw = eng.get_state("qwin"); h = eng.get_state("hist")
    nbad = 0
    for e in range(N):
        v = h[e][~np.isnan(h[e])]
        ref = eng.get_state("hist_ref")[e] if False else 0.0
        keys = np.sort(f32_key(v))
        n = len(keys)
        for side, base in ((0, 16), (1, 32), (2, 18), (3, 20)):
            r0, hi = int(hd[e, base].view(np.int32) if hasattr(hd[e, base], "view") else hd[e, base]), int(hd[e, base + 1])
            r0 = int(np.int32(np.uint32(hd[e, base])))
            w = qw[e, :, side].astype(np.uint64)
            ks = keys if side < 3 else np.sort((~keys) & 0xFFFFFFFF)
            want = ks[r0:r0 + hi] if r0 >= 0 else None
            ok = hi > 0 and r0 >= 0 and r0 + hi <= n and np.array_equal(w[:hi], want) and (w[hi:] == 0xFFFFFFFF).all()
            if not ok:
                nbad += 1
                if nbad <= 3:
                    k = (n - 1) // 4 if side == 0 else (3 * (n - 1)) // 4
                    print("t", t, "env", e, "side", side, "n", n, "k", k, "r0", r0, "hi", hi, "path", inf[e, 39])
                    if hi > 0 and r0 >= 0:
                        mism = np.nonzero(w[:hi] != want)[0] if want is not None and len(want) == hi else None
                        print("   first mismatches", None if mism is None else mism[:8], "w[:4]", w[:4], "want[:4]", None if want is None else want[:4])
                        print("   w[hi-3:hi+2]", w[max(hi - 3, 0):hi + 2], "want tail", None if want is None else want[-3:])
    print("t", t, "bad windows", nbad, "faults", np.unique(inf[:, L.INFO_IDX["fault"]]), "paths", np.bincount(inf[:, 39].astype(int)))
