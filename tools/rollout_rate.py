"""Measurement: sdc_rollout (K env-steps per launch, action sequence resident in HBM) against K calls of sdc_step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
N = int(os.environ.get("SDC_N", "4096"))
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=0)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (96, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
done = 0
while done < 10300:                       # fill the history rings
    k = min(96, eng.steps_to_episode_end())
    eng.rollout(pool[:k], want_info=False)
    done += k
for K in (1, 4, 16, 48, 96):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    steps = 0
    while steps < 1920:
        k = min(K, eng.steps_to_episode_end())
        eng.rollout(pool[:k])
        steps += k
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("rollout K=%-3d  us per env-step batch %.2f   Menv-steps/s %.1f" % (K, dt / steps * 1e6, N * steps / dt / 1e6))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(1920):
    eng.step(pool[i % 96])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("single steps    us per env-step batch %.2f   Menv-steps/s %.1f" % (dt / 1920 * 1e6, N * 1920 / dt / 1e6))
