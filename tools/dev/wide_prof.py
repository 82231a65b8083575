"""A few hundred steady-state step launches at N envs with full history rings (injected: no 10 000-step fill) for a profiler.
usage: python tools/dev/wide_prof.py N [steps] ; SDC_DBG = debug_flags (4096: lane-per-env kernel off)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
N = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
cap = 10000
tb = traces.synthetic_tables("ny", 0)
p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
rng = np.random.default_rng(3)
e = SdcEngine(N, episode_steps=672, auto_reset=True, seed=12, debug_flags=int(os.environ.get("SDC_DBG", "0")))
e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"]); e.set_dc_params(0, p); e.assign(0, 0, 174, 188)
base = (331 + 70 * rng.standard_normal((4096, cap))).clip(150, 650).astype(np.float32)
hist = np.full((N, 10240), np.nan, np.float32)
for i in range(0, N, 4096):
    hist[i:i + 4096, :cap] = base[: min(4096, N - i)] + np.float32(0.01 * (i // 4096))
e.set_state("hist", hist); e.set_state("hist_len", np.full(N, cap, np.int32)); e.set_state("hist_pos", rng.integers(0, cap, N).astype(np.int32))
del hist
e.reset()
g = torch.Generator(device="cpu").manual_seed(5)
acts = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).cuda()
for t in range(100): e.step(acts[t % 64])
torch.cuda.synchronize(); t0 = time.perf_counter()
for t in range(steps): e.step(acts[t % 64])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
rsv = e.info[:, 39]
print(f"N={N} dbg={os.environ.get('SDC_DBG','0')}: {dt / steps * 1e6:.2f} us/step {N * steps / dt / 1e6:.1f} M env-steps/s  paths", torch.unique(rsv, return_counts=True))
e.close()
