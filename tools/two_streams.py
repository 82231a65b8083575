"""Measurement: 4096 envs as G independent groups of 4096 / G envs, each group on its own HIP stream (its launches are
ordered, the groups are not): the launch gap, ramp and tail of one group's step overlap with the other groups' work.
That is how a double-buffered sampler (policy on one group while the other steps) would drive the engine."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
N = int(os.environ.get("SDC_N", "4096"))
G = int(os.environ.get("SDC_GROUPS", "2"))
n = N // G
engs, pools, streams = [], [], []
for gi in range(G):
    eng, tb, params = bench.build_engine(n, 672, 0, seed=1234 + gi, debug_flags=0)
    gen = torch.Generator(device="cpu").manual_seed(1234 + gi)
    pools.append(torch.randint(0, 3, (256, n, 3), dtype=torch.int32, generator=gen).to("cuda:0"))
    engs.append(eng)
    streams.append(torch.cuda.Stream())
    eng.use_stream(streams[-1])
for eng in engs: eng.reset()
torch.cuda.synchronize()
def run(k):
    for i in range(k):
        for gi in range(G):
            engs[gi].step(pools[gi][i & 255])
run(10300)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 2000
run(K)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("groups %d x %d envs: us per step of all %d envs %.2f  Menv-steps/s %.1f" % (G, n, N, dt / K * 1e6, N * K / dt / 1e6))
