import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
N = 4096
for P in (64, 1024, 4096):
    eng, tb, params = bench.build_engine(N, 672, 0, seed=1234)
    g = torch.Generator(device="cpu").manual_seed(1234)
    pool = torch.randint(0, 3, (P, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
    eng.reset()
    c = 0
    for i in range(10300): eng.step(pool[c % P]); c += 1
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(5000):
            eng.step(pool[c % P]); c += 1
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        inf = eng.info.cpu().numpy()
        print(f"pool {P}: {dt / 5000 * 1e6:.2f} us/step; refills last step {(inf[:,39]==1).sum()}")
    eng.close()
