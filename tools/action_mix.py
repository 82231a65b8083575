import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, numpy as np
N = 4096
def run(mode):
    eng, tb, params = bench.build_engine(N, 672, 0, seed=1234)
    g = torch.Generator(device="cpu").manual_seed(1234)
    pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
    if mode == "ls1": pool[:, :, 0] = 1
    if mode == "bat2": pool[:, :, 2] = 2
    if mode == "idle": pool[:, :, 0] = 1; pool[:, :, 1] = 1; pool[:, :, 2] = 2
    eng.reset()
    for i in range(10300): eng.step(pool[i & 255])
    eng.profile(8); eng.profile_read(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(1500): eng.step(pool[i & 255])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = eng.profile_read(reset=True)
    print("%-6s us/step %.2f kernel %.2f" % (mode, dt / 1500 * 1e6, p["dynamics_ms"] / max(1, p["steps"]) * 1e3))
    eng.close()
for m in ("rand", "ls1", "bat2", "idle"): run(m)
