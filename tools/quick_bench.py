"""Quick steady-state timing of the step kernel (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
N = int(os.environ.get("SDC_N", "4096"))
dbg = int(os.environ.get("SDC_DBG", "0"))
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=dbg)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300):
    eng.step(pool[i & 255])
eng.profile(8); eng.profile_read(reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(2000):
    eng.step(pool[i & 255])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
p = eng.profile_read(reset=True)
import numpy as np
paths = np.bincount(eng.info[:, 39].cpu().numpy().astype(int), minlength=4)
print("us/step %.2f  kernel us %.2f  Menv-steps/s %.1f  paths(last step) %s" % (dt / 2000 * 1e6, p["dynamics_ms"] / max(1, p["steps"]) * 1e3, N * 2000 / dt / 1e6, paths))
