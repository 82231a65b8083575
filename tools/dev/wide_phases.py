"""Per-phase wall-clock time of the lane-per-env kernel's wavefronts (a -DSDC_WIDE_STAMPS build: tools/dev/ab_wide.sh)."""
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
names = ["loads issued (first round trip)", "record / row read from LDS", "late loads + header read + key gathers (second round trip)", "load shifting",
         "set-point + rack classes", "rack slots (tree sums)", "hvac / battery / append", "arrivals (whole wavefront)", "outside tests + window updates (whole wavefront)",
         "resolve + bounds + tails + crossings", "moments + ahead tests", "requests (whole wavefront) + rewards + header commit", "state out: ring, queue table, record block",
         "header block out + fallback", "outputs"]
NS = len(names)
acc = []
for t in range(100):
    e.step(acts[t % 64]); acc.append(e.info[::64, :NS + 1].cpu().numpy().copy())
a = np.stack(acc)          # [steps, waves, 13]

d = a[..., :NS] / 100.0    # us (100 MHz clock)
print(f"N={N}: per wavefront, mean / p90 us by phase (100 launches x {a.shape[1]} wavefronts)")
for i, nm in enumerate(names): print(f"  {nm:58s} {d[..., i].mean():7.2f} {np.percentile(d[..., i], 90):7.2f}")
print(f"  {'total':58s} {d.sum(-1).mean():7.2f} {np.percentile(d.sum(-1), 90):7.2f}")
st = a[..., NS]
print("  wavefront start spread within a launch (us): p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile((st - st.min(1, keepdims=True)) % (1 << 20), [50, 90, 100]) / 100.0))
