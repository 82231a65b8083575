"""Timeline of the lane-per-env kernel's two wavefronts (a -DSDC_WIDE_STAMPS build: tools/dev/ab_wide.sh): wall-clock stamps of lane 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
N = int(sys.argv[1]); cap = 10000
tb = traces.synthetic_tables("ny", 0); p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
rng = np.random.default_rng(3)
e = SdcEngine(N, episode_steps=672, auto_reset=True, seed=12, debug_flags=2048)
e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"]); e.set_dc_params(0, p); e.assign(0, 0, 174, 188)
base = (331 + 70 * rng.standard_normal((4096, cap))).clip(150, 650).astype(np.float32)
hist = np.full((N, 10240), np.nan, np.float32)
for i in range(0, N, 4096): hist[i:i + 4096, :cap] = base[: min(4096, N - i)]
e.set_state("hist", hist); e.set_state("hist_len", np.full(N, cap, np.int32)); e.set_state("hist_pos", rng.integers(0, cap, N).astype(np.int32))
e.reset()
g = torch.Generator(device="cpu").manual_seed(5)
acts = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).cuda()
for t in range(150): e.step(acts[t % 64])
acc = []
for t in range(100):
    e.step(acts[t % 64]); acc.append(e.info[::64, :16].cpu().numpy().astype(np.int64).copy())
a = np.stack(acc)                                  # [steps, workgroups, 16] ticks mod 2^24
t0 = a[..., 0:1]
d = ((a - t0) % (1 << 24)) / 100.0                 # us after the dynamics wavefront's entry
names = ["D entry", "D first blocks in LDS (barrier 1)", "D load shifting done", "D rack model done", "D energy known (hand-over)", "D state out starts",
         "D outputs start", "D info staged (barrier 3)", "R headers in LDS (barrier 1)", "R arrivals done, at barrier 2", "R past barrier 2: updates start",
         "R window updates done", "R moments done", "R committed", "R header out + fallback done", "R info out"]
print(f"N={N}: mean / p90 us after the dynamics wavefront's entry (100 launches x {a.shape[1]} workgroups)")
for i, nm in enumerate(names): print(f"  {nm:48s} {d[..., i].mean():7.2f} {np.percentile(d[..., i], 90):7.2f}")
st = a[..., 0]
sp = ((st - st.min(1, keepdims=True)) % (1 << 24)) / 100.0
print("  workgroup entry spread within a launch (us): p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(sp, [50, 90, 100])))
en = ((a[..., 15] - st.min(1, keepdims=True)) % (1 << 24)) / 100.0
print("  last stamp after the launch's first entry (us): p50 %.2f p90 %.2f max(mean over launches) %.2f" % (np.percentile(en, 50), np.percentile(en, 90), en.max(1).mean()))
