import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
N = 512
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=2)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 63])
prev = None
shown = 0
for i in range(200):
    hd0 = eng.get_state("header"); qw0 = eng.get_state("qwin")
    o, s, r, d, info = eng.step(pool[i & 63])
    p = info[:, 39].cpu().numpy().astype(int)
    hd1 = eng.get_state("header")
    for e in np.nonzero((p == 9) | (p == 10))[0][:2]:
        if shown >= 6: break
        shown += 1
        side = 0 if p[e] == 9 else 1
        base = 18 if side == 0 else 20
        w = qw0[e, :, 2 + side]
        r0, hi = int(np.int32(hd0[e, base])), int(np.int32(hd0[e, base + 1]))
        print("step", i, "env", e, "code", p[e], "pre: r0", r0, "hi", hi, "n", int(hd0[e, 0]), "qc", int(np.int32(hd0[e, 29 + side])), "kb_last", int(hd0[e, 49 + side]),
              "w0", int(w[0]), "wtop", int(w[max(hi - 1, 0)]), "| post kb", int(hd1[e, 49 + side]), "post qc", int(np.int32(hd1[e, 29 + side])), "post r0/hi", int(np.int32(hd1[e, base])), int(np.int32(hd1[e, base + 1])))
