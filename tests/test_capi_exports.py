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


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors of the header's structs against the C compiler's view of include/sustaindc_hip.h: sizes and
    the offset of every member."""
    import subprocess
    members = {
        "sdc_config": ["n_envs", "device", "episode_steps", "hist_cap", "queue_max_len", "n_locations", "n_dc_configs",
                       "auto_reset", "seed", "weather_noise_std", "weather_noise_weight", "max_roll_days", "debug_flags",
                       "reward_method", "env_index_base", "policy", "reserved2", "trim_and_respond_limit"],
        "sdc_dc_params": ["n_racks", "rack_n", "rack_full", "rack_idle", "rack_supply", "rack_return", "m_cpu",
                          "itfan_ref_p", "c_air", "ct_fan_ref_p", "min_temp", "init_setpoint", "bat_capacity_mwh"],
        "sdc_reset_override": ["day", "hour", "ci_min", "ci_max", "t_min", "t_max", "t_win", "wb_win", "noise", "roll_days"],
    }
    mirror = {"sdc_config": L.SdcConfig, "sdc_dc_params": L.SdcDcParams, "sdc_reset_override": L.SdcResetOverride}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "sustaindc_hip.h")}"',
           "int main(void) {"]
    for st, ms in members.items():
        src.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for m in ms:
            src.append(f'  printf("{st}.{m} %zu\\n", offsetof({st}, {m}));')
    src += ["  return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-o", exe, str(c)], check=True)
    out = dict(ln.split() for ln in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines())
    for st, ms in members.items():
        assert int(out[st]) == C.sizeof(mirror[st]), st
        for m in ms:
            assert int(out[f"{st}.{m}"]) == getattr(mirror[st], m).offset, (st, m)
    assert [f[0] for f in L.SdcConfig._fields_] == members["sdc_config"]     # every member of sdc_config is mirrored
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
