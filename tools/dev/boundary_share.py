"""What the episode boundary (sdc_reset_kernel + sdc_features_kernel, once per episode) costs per step at batch size N: the step inside
an episode (600 launches between two boundaries) against whole episodes (3 x 672 steps, auto-resets inside).
usage: python tools/dev/boundary_share.py N [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
for N in [int(x) for x in sys.argv[1:]] or (4096, 32768):
    eng, tb, params = bench.build_engine(N, 672, 0, seed=1234)
    g = torch.Generator(device="cpu").manual_seed(1234)
    P = 256
    pool = torch.randint(0, 3, (P, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
    eng.reset()
    c = 0
    for i in range(672 * 15 + 20):          # 10 100 steps: rings full, 20 steps into an episode
        eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(600):
        eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); t_in = (time.perf_counter() - t0) / 600
    for i in range(672 - 620):
        eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3 * 672):
        eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / (3 * 672)
    print(f"N={N} {eng.last_step_kernel()}: inside an episode {t_in * 1e6:.2f} us per step, whole episodes {t_all * 1e6:.2f} "
          f"-> boundary {(t_all - t_in) * 1e6:.2f} us per step amortised = {(t_all - t_in) * 672 * 1e6:.0f} us per episode "
          f"({(t_all - t_in) / t_all * 100:.1f} % of the rate)", flush=True)
    eng.close()
