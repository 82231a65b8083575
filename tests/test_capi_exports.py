"""The C-ABI shared library loads and exports every symbol include/sustaindc_hip.h declares (no compute calls)."""
import ctypes as C
import os
import re

from dc_rl_amd import _lib as L

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sustaindc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdc_[a-z_]+)\s*\(", src)))


def test_library_builds_and_exports_all_declared_symbols():
    L.build()
    lib = C.CDLL(L.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in sustaindc_hip.h but not exported"
    assert set(names) == set(L.EXPORTS)


def test_struct_sizes_match_header():
    # sdc_dc_params: 2 x int32 + 5 x 64 doubles + 18 doubles
    assert C.sizeof(L.SdcDcParams) == 8 + 8 * (5 * 64 + 18)
    # sdc_config: 8 x int32, uint64, 2 doubles, 2 x int32, reward_method[3] + reserved
    assert C.sizeof(L.SdcConfig) == 8 * 4 + 8 + 8 + 8 + 8 + 16
    assert len(L.INFO_COLS) == L.INFO_DIM


def test_fails_loudly_without_gpu_or_bad_args():
    import torch
    lib = L.load()
    h = C.c_void_p()
    cfg = L.SdcConfig(n_envs=0, device=0, episode_steps=672, hist_cap=10000, queue_max_len=1000, n_locations=1,
                      n_dc_configs=1, auto_reset=1, seed=0, weather_noise_std=0.75, weather_noise_weight=0.02,
                      max_roll_days=14, debug_flags=0)
    assert lib.sdc_create(C.byref(cfg), C.byref(h)) != 0
    assert b"n_envs" in lib.sdc_last_error()
    if not torch.cuda.is_available():
        cfg.n_envs = 4
        assert lib.sdc_create(C.byref(cfg), C.byref(h)) != 0  # no device: error, never a CPU fallback
        import pytest
        from dc_rl_amd.engine import SdcEngine
        with pytest.raises(RuntimeError):
            SdcEngine(4)
