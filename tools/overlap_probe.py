"""How much do reset + feature kernels running on a SECOND stream slow the step kernel down?  Engine A steps 4096 envs on its
stream; engine B (same size) is reset every `every` A-steps on another stream (a reset = 0.55 ms of GPU work = ~46 A-steps).
usage: python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
N = 4096
A, _, _ = bench.build_engine(N, 672, 0, seed=1)
B, _, _ = bench.build_engine(N, 672, 0, seed=2)
sa, sb = torch.cuda.Stream(priority=int(os.environ.get('SA_PRIO', '0'))), torch.cuda.Stream(priority=0)
try:
    lo, hi = torch.cuda.Stream.priority_range()
except Exception:
    lo, hi = 0, -1
sb_low = torch.cuda.Stream(priority=lo)
A.use_stream(sa)
g = torch.Generator(device="cuda").manual_seed(1)
pool = torch.randint(0, 3, (512, N, 3), dtype=torch.int32, device="cuda", generator=g)
A.reset(); B.use_stream(sb); B.reset()
torch.cuda.synchronize()
k = 0
for i in range(10300):
    A.step(pool[k % 512]); k += 1
torch.cuda.synchronize()
def run(every, stream, steps=2600):
    global k
    B.use_stream(stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        if A.steps_to_episode_end() <= 1:
            pass
        A.step(pool[k % 512]); k += 1
        if every and i % every == 0:
            B.reset()
    sa.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / steps * 1e6
print("priority range", lo, hi)
for rep in range(2):
    print("alone            %.2f us/step" % run(0, sb))
    for every in (46, 336, 672):
        print("B reset every %3d (normal prio) %.2f   (low prio stream) %.2f" % (every, run(every, sb), run(every, sb_low)))
