import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
for N in [int(x) for x in sys.argv[1:]] or (2048, 4096, 8192, 16384):
    eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=int(os.environ.get("SDC_DBG", "0")))   # (512: two envs per wavefront, 1024: four)
    g = torch.Generator(device="cpu").manual_seed(1234)
    P = 256
    pool = torch.randint(0, 3, (P, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
    eng.reset()
    c = 0
    for i in range(10300): eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3000):
        eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={N}: {dt / 3000 * 1e6:.2f} us/step  {N * 3000 / dt / 1e6:.1f} M env-steps/s", flush=True)
    eng.close()
