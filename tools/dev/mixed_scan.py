"""Step time of a batch of N envs, one DC config or BASELINE configs[3]'s 16 / 20 / 25-rack mix (--mixed), rings filled by real
steps, i.i.d. device-resident actions, auto-resets inside the timed region; names the kernel the host picked.
usage: python tools/dev/mixed_scan.py [--mixed] [--fill 10300] [--steps 3000] N [N ...]   (SDC_DBG = debug_flags)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench

ap = argparse.ArgumentParser()
ap.add_argument("--mixed", action="store_true")
ap.add_argument("--fill", type=int, default=10300)
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("sizes", nargs="*", type=int, default=[32768])
a = ap.parse_args()
files = ("dc_config.json", "dc_config_r16.json", "dc_config_r25.json") if a.mixed else ("dc_config.json",)
for N in a.sizes:
    eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=int(os.environ.get("SDC_DBG", "0")), dc_files=files)
    g = torch.Generator(device="cpu").manual_seed(1234)
    P = 256
    pool = torch.randint(0, 3, (P, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
    eng.reset()
    c = 0
    for i in range(a.fill):
        eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(a.steps):
        eng.step(pool[c % P]); c += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"N={N} mixed={a.mixed} {eng.last_step_kernel()}: {dt / a.steps * 1e6:.2f} us/step  {N * a.steps / dt / 1e6:.1f} M env-steps/s "
          f"faults={int((eng.info[:, 37] != 0).sum())}", flush=True)
    eng.close()
