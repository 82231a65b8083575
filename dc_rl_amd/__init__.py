"""dc_rl_amd (also reachable as the directory name `dc-rl_amd`, a symlink): MI355X-native vectorised SustainDC step behind the reference's own surface.

Public surface (mirrors the reference's names for this path):
  * `SustainDC`, `EnvConfig`                      -- sustaindc_env.py:34, :90
  * `make_ls_env`, `make_dc_pyeplus_env`, `make_bat_fwd_env` -- utils/make_envs_pyenv.py:19, :75, :45
  * `SustainDCVecEnv` (a `ShareVecEnv`), `make_train_env`, `make_eval_env`, `make_render_env`
                                                   -- harl/envs/env_wrappers.py:53, harl/utils/envs_tools.py:49, :77, :106
  * `install_into_harl()`                          -- rebinds those factories inside an UNCHANGED HARL tree (zero edits)
  * `SustainDCMultiDeviceVecEnv` / `make_train_env(..., devices=[...])` -- the same ShareVecEnv over several GPUs in one process
  * `SdcEngine`                                    -- thin ctypes wrapper over the C-ABI (include/sustaindc_hip.h)

The compute path is the HIP extension `csrc/libsustaindc_hip.so` (hand-written gfx950 kernels).  There is
no CPU fallback: constructing an engine without the extension or without an MI355X raises.
"""
__version__ = "0.1.0"

_LAZY = {
    "SdcEngine": ("engine", "SdcEngine"),
    "SustainDC": ("sustaindc_env", "SustainDC"),
    "EnvConfig": ("sustaindc_env", "EnvConfig"),
    "SustainDCVecEnv": ("vec_env", "SustainDCVecEnv"),
    "ShareVecEnv": ("vec_env", "ShareVecEnv"),
    "SustainDCMultiDeviceVecEnv": ("multi_device", "SustainDCMultiDeviceVecEnv"),
    "make_train_env": ("envs_tools", "make_train_env"),
    "make_eval_env": ("envs_tools", "make_eval_env"),
    "make_render_env": ("envs_tools", "make_render_env"),
    "install_into_harl": ("envs_tools", "install_into_harl"),
    "make_ls_env": ("make_envs_pyenv", "make_ls_env"),
    "make_dc_pyeplus_env": ("make_envs_pyenv", "make_dc_pyeplus_env"),
    "make_bat_fwd_env": ("make_envs_pyenv", "make_bat_fwd_env"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module("dc_rl_amd." + mod), attr)
    raise AttributeError(name)
