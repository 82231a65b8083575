import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
N, steps, cap = 1024, 96, 10000
tb = traces.synthetic_tables("ny", 0)
p = dc_config.size_datacenter("dc_config.json", 1, 30.0)
rng = np.random.default_rng(3)
hist = np.full((N, 10240), np.nan, np.float32)
hist[:, :cap] = (331 + 70 * rng.standard_normal((N, cap))).clip(150, 650).astype(np.float32)
pos = rng.integers(0, cap, N).astype(np.int32)
engs = []
for flags in (2048, 512):
    e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=12, debug_flags=flags)
    e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"]); e.set_dc_params(0, p); e.assign(0, 0, 174, 188)
    e.set_state("hist", hist); e.set_state("hist_len", np.full(N, cap, np.int32)); e.set_state("hist_pos", pos)
    e.reset(); engs.append(e)
a, b = engs
g = torch.Generator(device="cpu").manual_seed(5)
acts = torch.randint(0, 3, (300, N, 3), dtype=torch.int32, generator=g).cuda()
rsv = L.INFO_IDX["reserved"]
shown = 0
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    a.step(acts[t]); b.step(acts[t])
    ha, hb = a.get_state("header"), b.get_state("header")
    ha[:, 34:38] &= ~np.uint32(0x7FF); hb[:, 34:38] &= ~np.uint32(0x7FF)
    bad = np.argwhere(ha != hb)
    ra, rb = a.info[:, rsv].cpu().numpy(), b.info[:, rsv].cpu().numpy()
    print("step", t, "hdr diffs", len(bad), "envs", len(set(bad[:, 0].tolist())), "reserved wide", np.unique(ra, return_counts=True), "pair", np.unique(rb, return_counts=True))
    for e in sorted(set(bad[:, 0].tolist()))[:3]:
        if shown < 12:
            shown += 1
            d = [int(j) for j in bad[bad[:, 0] == e][:, 1]]
            print("   env", e, "dwords", d, "wide", [int(ha[e, j]) for j in d], "pair", [int(hb[e, j]) for j in d], "rsv", ra[e], rb[e],
                  "pend wide", [hex(int(x)) for x in ha[e, 34:38]], "pair", [hex(int(x)) for x in hb[e, 34:38]])
