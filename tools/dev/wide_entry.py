"""-DSDC_WIDE_STAMPS build: the lane-per-env kernel's workgroups by XCD class (b % 8): the workgroup's FIRST instruction (before it has
read a kernel argument), the dynamics wavefront's entry stamp (after the arguments), blocks in LDS, end -- after the launch's first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
N = int(sys.argv[1])
e, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=2048)
g = torch.Generator(device="cpu").manual_seed(1234)
acts = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
e.reset()
for t in range(int(os.environ.get("FILL", "10300"))): e.step(acts[t % 64])
acc = []
for t in range(100):
    e.step(acts[t % 64]); acc.append(e.info[::64, :17].cpu().numpy().astype(np.int64).copy())
a = np.stack(acc)                                  # [steps, env blocks, 17]
nb = a.shape[1]
top = a[..., 16]
first = top.min(1, keepdims=True)
rel = lambda x: ((x - first) % (1 << 24)) / 100.0
cls = np.arange(nb) // (nb // 8)                   # env block -> XCD class (first_pair_of_block)
print(f"N={N}: us after the launch's first workgroup's first instruction, mean by XCD class 0..7")
for nm, col in (("first instruction", 16), ("D entry (arguments read)", 0), ("blocks in LDS", 1), ("energy handed over", 4), ("R info out (end)", 15)):
    r = rel(a[..., col])
    print(f"  {nm:28s}", " ".join("%6.2f" % r[:, cls == c].mean() for c in range(8)), "| all p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(r, [50, 90, 100])))
