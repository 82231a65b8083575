"""Where sdc_reset_kernel's time goes: phase stamps of a MEASUREMENT build (-DSDC_RT: lane 0 of every wavefront writes the
wall clock at eight points into the tail of its env's queue table; the production build has none of it).

usage (GPU box, the library built with -DSDC_RT, e.g. through tools/ab.sh):  python tools/reset_phases.py [n_envs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dc_rl_amd import dc_config, traces  # noqa: E402
from dc_rl_amd.engine import SdcEngine  # noqa: E402

NAMES = ["record + draws", "noise walk", "walk moments", "pass 2: add / roll / clip / bounds", "bounds reductions",
         "obs windows + obs pool", "record, outputs"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    tb = traces.synthetic_tables("ny", 0)
    eng = SdcEngine(n, episode_steps=672, auto_reset=True, seed=1)
    eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
    eng.set_dc_params(0, dc_config.size_datacenter("dc_config.json", 1, 30.0))
    eng.assign(0, 0, 170, 190)
    for _ in range(4):
        eng.reset()
    q = eng.get_state("qtab")
    st = np.ascontiguousarray(q.reshape(n, -1)[:, -16:]).view(np.uint64).astype(np.float64)     # [n, 8] stamps
    tick_us = 0.01      # 100 MHz constant-rate clock
    t0 = st[:, 0].min()
    print(f"{n} envs: first entry -> last exit {(st[:, 7].max() - t0) * tick_us:.1f} us; entry p50 {np.median(st[:, 0] - t0) * tick_us:.1f}"
          f" max {(st[:, 0].max() - t0) * tick_us:.1f}")
    for i, name in enumerate(NAMES):
        d = (st[:, i + 1] - st[:, i]) * tick_us
        print(f"  {name:38s} mean {d.mean():7.1f}  p50 {np.median(d):7.1f}  max {d.max():7.1f} us")
    hw = q.reshape(n, -1, 2)[:, -9, 0].astype(np.int64)      # XCC << 16 | HW_ID: wave [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
    simd = (hw >> 4) & 0xFFFFF
    per = np.bincount(np.unique(simd, return_inverse=True)[1])
    print(f"  wavefronts per SIMD: {len(per)} SIMDs used, min {per.min()} max {per.max()}, histogram {np.bincount(per).tolist()}")
    life = (st[:, 7] - st[:, 0]) * tick_us
    print(f"  wavefront lifetime mean {life.mean():.1f} p50 {np.median(life):.1f} max {life.max():.1f} us")


if __name__ == "__main__":
    main()
