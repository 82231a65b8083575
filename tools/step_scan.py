"""us per sdc_step launch (no episode boundary inside the timed window) by batch size and lane mapping (development aid):
debug_flags 512 = two envs per wavefront, 1024 = four."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
for N in [int(x) for x in os.environ.get("SDC_NS", "4096,8192,12288,16384").split(",")]:
    for flags in (512, 1024):
        eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=flags)
        g = torch.Generator(device="cuda").manual_seed(1234)
        pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, device="cuda", generator=g)
        eng.reset()
        for i in range(120):
            eng.step(pool[i % 256])
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(150):
                eng.step(pool[(i + 7 * rep) % 256])
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 150 * 1e6)
        done = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        while done < 384:
            kk = min(48, eng.steps_to_episode_end())
            eng.rollout(pool[:kk]); done += kk
        torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / done * 1e6
        print("N %6d  %s: %.2f us per step = %.1f M env-steps/s;  sdc_rollout (48 per launch) %.2f us per step = %.1f M" % (
            N, "two envs per wavefront " if flags == 512 else "four envs per wavefront", best, N / best, tr, N / tr))
        eng.close()
