"""The lane-per-env step kernel (csrc/sdc_wide.hip, the largest batches; forced here by debug_flags bit 11) against the two-envs-per-
wavefront kernel (bit 9): same arithmetic in the same order, so every output and the state are the same BITS."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine

pytestmark = pytest.mark.gpu

WIDE, PAIR = 2048, 512


def _engines(N, steps, cfg="dc_config.json", seed=3, days=(200, 210), flags=(WIDE, PAIR)):
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter(cfg, 1, 30.0)
    engs = []
    for fl in flags:
        e = SdcEngine(N, episode_steps=steps, auto_reset=True, seed=seed, debug_flags=fl)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        e.assign(0, 0, *days)
        e.reset()
        engs.append(e)
    return engs


def _same_step(a, b, acts, t, what, skip_reserved=True):
    import torch
    rsv = L.INFO_IDX["reserved"]
    for u, v, nm in zip(a.step(acts), b.step(acts), ("obs", "share_obs", "rew", "done", "info")):
        if nm == "info" and skip_reserved:
            u, v = u.clone(), v.clone()
            u[:, rsv] = 0
            v[:, rsv] = 0
        if not torch.equal(u, v):
            bad = (u != v).nonzero()
            raise AssertionError((what, t, nm, bad[:6].tolist(), u[tuple(bad[0])].item(), v[tuple(bad[0])].item()))


@pytest.mark.parametrize("cfg", ["dc_config.json", "dc_config_r16.json", "dc_config_r25.json"])
def test_lane_per_env_kernel_equals_the_pair_kernel(cfg):
    """256 envs (four wavefronts), 96-step episodes, 230 steps = two auto-resets; 20 / 16 / 25 racks (the rack sums' tree has a
    different shape for each)."""
    import torch
    N, steps = 256, 96
    a, b = _engines(N, steps, cfg)
    g = torch.Generator(device="cpu").manual_seed(9)
    acts = torch.randint(0, 3, (230, N, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(230):
        _same_step(a, b, acts[t], t, cfg)
    for name in ("record", "hist", "qtab"):
        np.testing.assert_array_equal(a.get_state(name), b.get_state(name), err_msg=name)
    assert (a.info[:, L.INFO_IDX["fault"]] == 0).all()
    a.close()
    b.close()
