"""Soak of the lane-per-env kernel in VERIFY mode (debug_flags bit 0: after every step sdc_reward_verify_kernel checks every key of all
four rank windows against its rank in the ring, the quartiles against an exact bisection, z against a direct fp64 pass): N envs from
empty rings through the 10 000-step fill (young histories, the first re-centrings, the request flood around step 64-100) and four more
episodes.   usage: python tools/dev/wide_soak.py [N] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
from dc_rl_amd import _lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12700
eng, tb, params = bench.build_engine(N, 672, 0, seed=4321, debug_flags=1)
g = torch.Generator(device="cpu").manual_seed(77)
pool = torch.randint(0, 3, (128, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
paths = {}
for i in range(steps):
    o, s, r, d, info = eng.step(pool[i & 127])
    if i % 50 == 49 or i == steps - 1:
        f = info[:, L.INFO_IDX["fault"]]
        assert (f == 0).all(), (i, torch.unique(f))
        assert torch.isfinite(r).all(), i
        u, c = torch.unique(info[:, L.INFO_IDX["reserved"]], return_counts=True)
        for a, b in zip(u.tolist(), c.tolist()): paths[int(a)] = paths.get(int(a), 0) + b
assert eng.last_step_kernel() == "sdc_dynamics_wide_kernel", eng.last_step_kernel()
st = eng.get_state("order_stat_sticky")
print(f"N={N} steps={steps}: faults 0, verify-mode mismatches {int((st != 0).sum())}, reward paths sampled every 50 steps {paths}")
assert (st == 0).all()
