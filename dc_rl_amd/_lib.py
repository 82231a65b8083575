"""ctypes binding of the C-ABI in include/sustaindc_hip.h (csrc/libsustaindc_hip.so).

Loading fails loudly: there is no CPU fallback for the product path.  `import torch` must happen
before the library is loaded so that the HIP runtime already mapped into the process (PyTorch-ROCm's
libamdhip64.so.7) is the one our kernels launch through -- tensors and kernels then share one context.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

MAX_RACKS = 64
N_AGENTS = 3
OBS_PAD = 26
SHARE_OBS_DIM = 29
INFO_DIM = 44
TABLE_LEN = 35040
HDR_DWORDS = 64    # csrc/sdc_device.hpp SdcHdr: 256-byte per-env header
QWIN = 64          # csrc/sdc_device.hpp SDC_WIN: keys per rank window (4 windows per env: Q1, Q3, upper / lower clip bound)
ABI_VERSION = 311  # include/sustaindc_hip.h SDC_ABI_VERSION: the struct layouts and argument lists this binding was written for

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(CSRC, "libsustaindc_hip.so")
SOURCES = ["sdc_capi.hip", "sdc_step.hip", "sdc_rollout.hip", "sdc_wide.hip", "sdc_features.hip", "sdc_verify.hip", "sdc_reset.hip"]
# (-amdgpu-sched-strategy=max-ilp: the machine scheduler orders for instruction-level parallelism instead of minimal register
#  pressure -- the step kernels' occupancy is pinned by amdgpu_waves_per_eu anyway, and their time is dependent-issue latency:
#  measured 12.09 -> 11.78 us per step of 4096 envs, the large-batch kernels +1-2 %)
# (-disable-machine-licm, round 4: the multi-step kernels run the whole step inside a loop, and the machine-level
#  loop-invariant code motion hoists every constant materialisation of the step -- LDS offsets, fp64 literal halves, cold-path
#  constants -- in front of it, holding 20-50 VGPRs across the loop: sdc_rollout_kernel 218 -> 167 VGPRs, the closed-loop
#  kernels 177 / 185 -> 151 / 159, and sdc_rollout_quad_kernel's 32 bytes of scratch are gone (tests/test_isa_guard.py);
#  measured: single step unchanged, sdc_rollout 8.55 -> 8.46 us per step at 4096 envs, 25.7 -> 23.8 at 16 384)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-disable-machine-licm"]

# info column names = the reference's info keys (sustaindc_hip.h enum sdc_info_col)
INFO_COLS = [
    "ls_original_workload", "ls_shifted_workload", "ls_tasks_in_queue", "ls_norm_tasks_in_queue",
    "ls_tasks_dropped", "ls_tasks_processed", "ls_oldest_task_age", "ls_average_task_age",
    "ls_overdue_penalty", "ls_computed_tasks", "ls_current_hour",
    "ls_task_age_hist0", "ls_task_age_hist1", "ls_task_age_hist2", "ls_task_age_hist3", "ls_task_age_hist4",
    "dc_ITE_total_power_kW", "dc_CT_total_power_kW", "dc_Compressor_total_power_kW",
    "dc_HVAC_total_power_kW", "dc_total_power_kW", "dc_crac_setpoint_delta", "dc_crac_setpoint",
    "dc_cpu_workload_fraction", "dc_int_temperature", "dc_exterior_ambient_temp", "dc_water_usage",
    "bat_action", "bat_SOC", "bat_CO2_footprint", "bat_avg_CI",
    "bat_total_energy_without_battery_KWh", "bat_total_energy_with_battery_KWh",
    "norm_CI", "outside_temp", "day", "hour", "fault", "energy_z", "reserved",
    "ep_return_ls", "ep_return_dc", "ep_return_bat", "episode_step",
]
INFO_IDX = {k: i for i, k in enumerate(INFO_COLS)}
assert len(INFO_COLS) == INFO_DIM


class SdcConfig(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("device", C.c_int32), ("episode_steps", C.c_int32), ("hist_cap", C.c_int32),
        ("queue_max_len", C.c_int32), ("n_locations", C.c_int32), ("n_dc_configs", C.c_int32),
        ("auto_reset", C.c_int32), ("seed", C.c_uint64), ("weather_noise_std", C.c_double),
        ("weather_noise_weight", C.c_double), ("max_roll_days", C.c_int32), ("debug_flags", C.c_int32),
        ("reward_method", C.c_int32 * 3), ("env_index_base", C.c_int32),
        ("policy", C.c_int32 * 3), ("reserved2", C.c_int32), ("trim_and_respond_limit", C.c_double),
    ]


# reward-method codes (include/sustaindc_hip.h enum sdc_reward_method) by agent slot and reference name
# (utils/reward_creator.py:322-334).  default_ls_reward is only available to agent_ls: each call appends to the
# shared energy history (reward_creator.py:63).
_ALT_REWARDS = {"custom_agent_reward": 2, "tou_reward": 3, "energy_efficiency_reward": 4, "energy_PUE_reward": 5,
                "water_usage_efficiency_reward": 6}
REWARD_CODES = {
    "ls_reward": {"default_ls_reward": 0, "default_dc_reward": 1, "default_bat_reward": 1, **_ALT_REWARDS},
    "dc_reward": {"default_dc_reward": 0, "default_bat_reward": 1, **_ALT_REWARDS},
    "bat_reward": {"default_bat_reward": 0, "default_dc_reward": 1, **_ALT_REWARDS},
}


def reward_codes(env_args: dict):
    """(ls, dc, bat) reward-method codes of an env_config; NotImplementedError for what the device does not run."""
    out = []
    for key, table in REWARD_CODES.items():
        name = env_args.get(key, {"ls_reward": "default_ls_reward", "dc_reward": "default_dc_reward",
                                  "bat_reward": "default_bat_reward"}[key])
        if name not in table:
            raise NotImplementedError(
                f"{key}={name!r}: runs on the device: {sorted(table)} (renewable_energy_reward and "
                "temperature_efficiency_reward need inputs the reference env never provides; default_ls_reward is "
                "only available to agent_ls)")
        out.append(table[name])
    return tuple(out)


class SdcDcParams(C.Structure):
    _fields_ = [
        ("n_racks", C.c_int32), ("reserved", C.c_int32),
        ("rack_n", C.c_double * MAX_RACKS), ("rack_full", C.c_double * MAX_RACKS),
        ("rack_idle", C.c_double * MAX_RACKS), ("rack_supply", C.c_double * MAX_RACKS),
        ("rack_return", C.c_double * MAX_RACKS),
        ("m_cpu", C.c_double), ("c_cpu", C.c_double), ("rs_cpu", C.c_double),
        ("m_fan", C.c_double), ("c_fan", C.c_double), ("rs_fan", C.c_double),
        ("itfan_ref_p", C.c_double), ("itfan_ref_v_ratio", C.c_double), ("it_fan_full_load_v", C.c_double),
        ("c_air", C.c_double), ("rho_air", C.c_double), ("crac_supply_pu", C.c_double),
        ("ct_fan_ref_p", C.c_double), ("ctafr", C.c_double),
        ("min_temp", C.c_double), ("max_temp", C.c_double), ("init_setpoint", C.c_double),
        ("bat_capacity_mwh", C.c_double),
    ]


class SdcResetOverride(C.Structure):
    _fields_ = [
        ("day", C.POINTER(C.c_int32)), ("hour", C.POINTER(C.c_int32)),
        ("ci_min", C.POINTER(C.c_double)), ("ci_max", C.POINTER(C.c_double)),
        ("t_min", C.POINTER(C.c_double)), ("t_max", C.POINTER(C.c_double)),
        ("t_win", C.POINTER(C.c_double)), ("wb_win", C.POINTER(C.c_double)),
        ("noise", C.POINTER(C.c_double)), ("roll_days", C.POINTER(C.c_int32)),
    ]


class SdcActorParams(C.Structure):
    """One agent's actor network in torch's layout (include/sustaindc_hip.h sdc_actor_params)."""
    _fields_ = [
        ("ln0_gamma", C.c_float * 26), ("ln0_beta", C.c_float * 26),
        ("w1", C.c_float * (64 * 26)), ("b1", C.c_float * 64), ("ln1_gamma", C.c_float * 64), ("ln1_beta", C.c_float * 64),
        ("w2", C.c_float * (64 * 64)), ("b2", C.c_float * 64), ("ln2_gamma", C.c_float * 64), ("ln2_beta", C.c_float * 64),
        ("w3", C.c_float * (3 * 64)), ("b3", C.c_float * 3),
        ("use_feature_normalization", C.c_int32), ("activation", C.c_int32),
    ]


EXPORTS = [
    "sdc_last_error", "sdc_version", "sdc_create", "sdc_destroy", "sdc_set_seed", "sdc_weather_window_len", "sdc_set_tables",
    "sdc_set_dc_params", "sdc_assign_envs", "sdc_reset", "sdc_step", "sdc_rollout", "sdc_steps_to_episode_end",
    "sdc_last_done", "sdc_last_step_kernel",
    "sdc_get_state", "sdc_set_state",
    "sdc_hist_stride", "sdc_queue_stride", "sdc_profile_enable", "sdc_profile_read",
    "sdc_set_actor", "sdc_rollout_actor",
]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]   # (sdc_infos.c: build_infos)
    deps.append(os.path.join(CSRC, "..", "..", "include", "sustaindc_hip.h"))
    deps.append(os.path.abspath(__file__))      # (the compiler flags live here)
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(d) for d in deps):
        build_infos()
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # one object per translation unit, compiled side by side (the step kernels are ~35 s of compiler time, most of it three
    # files), then one link; objects of unchanged sources are reused (their own headers and the flags are their dependencies)
    extra = os.environ.get("SDC_HIPCC_EXTRA", "").split()      # (experiments: -D...)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + extra
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "flags.txt")
    # (the compiler is part of the stamp: objects of another hipcc are not reused)
    try:
        cc_id = subprocess.run([hipcc, "--version"], capture_output=True, text=True, check=True).stdout.strip().replace("\n", " | ")
    except Exception:
        cc_id = "unknown"
    flag_text = hipcc + " [" + cc_id + "] " + " ".join(cflags)
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True
    hdr_time = max(os.path.getmtime(d) for d in deps if not d.endswith(".hip"))

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return obj
        # (several ranks / test workers may build at once: each writes its own file, the rename is atomic)
        tmp_obj = obj + f".{os.getpid()}.tmp"
        cmd = [hipcc] + cflags + ["-c", src, "-o", tmp_obj]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.check_call(cmd, cwd=CSRC)
            os.replace(tmp_obj, obj)
        finally:
            if os.path.exists(tmp_obj):
                os.remove(tmp_obj)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    open(stamp, "w").write(flag_text)
    tmp = LIB_PATH + f".{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    try:
        build_infos(force)      # (host-side helper of the vector env's `infos`; an engine-only user does not need it)
    except Exception as e:
        print(f"dc_rl_amd: {INFOS_LIB} not built ({e!r}); SdcEngine works without it, SustainDCVecEnv needs it")
    return LIB_PATH


# ---- the host-side helper behind SustainDCVecEnv's `infos` (csrc/sdc_infos.c: CPython C API, gcc, no GPU code) ----------
INFOS_SRC = os.path.join(CSRC, "sdc_infos.c")
# (named with the interpreter's ABI tag -- _sdc_infos.cpython-310-x86_64-linux-gnu.so -- so that another Python version builds
# its own instead of importing this one)
import sysconfig as _sysconfig
INFOS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_sdc_infos" + (_sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_infos(force: bool = False) -> str:
    """Compile dc_rl_amd/_sdc_infos.so (the C types of `infos`) in-tree.  One second of gcc."""
    import sysconfig
    if not force and os.path.exists(INFOS_LIB) and os.path.getmtime(INFOS_LIB) >= os.path.getmtime(INFOS_SRC):
        return INFOS_LIB
    cc = os.environ.get("CC", "gcc")
    tmp = INFOS_LIB + f".{os.getpid()}.tmp"      # (several ranks / test workers may build at once: rename is atomic)
    subprocess.check_call([cc, "-O2", "-Wall", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], INFOS_SRC, "-o", tmp])
    os.replace(tmp, INFOS_LIB)
    untagged = os.path.join(os.path.dirname(INFOS_LIB), "_sdc_infos.so")
    if untagged != INFOS_LIB and os.path.exists(untagged):      # (a build of an earlier revision: the import system would still find it)
        os.remove(untagged)
    return INFOS_LIB


_infos_mod = None


def load_infos():
    """The `_sdc_infos` extension module (built on first use when it is missing or older than its source)."""
    global _infos_mod
    if _infos_mod is None:
        try:
            build_infos()
        except Exception as e:
            if not os.path.exists(INFOS_LIB):
                raise RuntimeError(f"{INFOS_LIB} is missing and could not be built ({e!r}: it needs gcc and the Python headers, "
                                   "one second); run `python -c 'import __graft_entry__ as g; g.build()'` on a machine that has "
                                   "them.  dc_rl_amd.SdcEngine (the C-ABI wrapper) works without it.") from e
        import importlib
        _infos_mod = importlib.import_module("dc_rl_amd._sdc_infos")
    return _infos_mod


_lib = None


def load():
    """Return the loaded library; raises RuntimeError if the extension is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (maps PyTorch-ROCm's HIP runtime first; see module docstring)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or dc_rl_amd._lib.build()). There is no CPU fallback for the SustainDC step.")
    L = C.CDLL(LIB_PATH)
    vp, ip, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_uint8)
    L.sdc_last_error.restype = C.c_char_p
    L.sdc_version.restype = C.c_int
    L.sdc_create.argtypes = [C.POINTER(SdcConfig), C.POINTER(vp)]
    L.sdc_destroy.argtypes = [vp]
    L.sdc_set_seed.argtypes = [vp, C.c_uint64]
    L.sdc_weather_window_len.argtypes = [vp]
    L.sdc_hist_stride.argtypes = [vp]
    L.sdc_queue_stride.argtypes = [vp]
    L.sdc_set_tables.argtypes = [vp, C.c_int, dp, dp, dp, dp, C.c_int]
    L.sdc_set_dc_params.argtypes = [vp, C.c_int, C.POINTER(SdcDcParams)]
    L.sdc_assign_envs.argtypes = [vp, ip, ip, ip, ip]
    L.sdc_reset.argtypes = [vp, u8p, C.POINTER(SdcResetOverride), fp, fp, vp]
    L.sdc_step.argtypes = [vp, vp, fp, fp, fp, vp, fp, fp, vp]
    L.sdc_rollout.argtypes = [vp, C.c_int, vp, fp, fp, fp, vp, fp, fp, vp, vp]
    L.sdc_steps_to_episode_end.argtypes = [vp]
    L.sdc_last_done.argtypes = [vp, u8p]
    L.sdc_last_step_kernel.argtypes = [vp]
    L.sdc_last_step_kernel.restype = C.c_char_p
    L.sdc_get_state.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    L.sdc_set_state.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    L.sdc_profile_enable.argtypes = [vp, C.c_int]
    L.sdc_profile_read.argtypes = [vp, dp, C.c_int]
    L.sdc_set_actor.argtypes = [vp, C.c_int, C.POINTER(SdcActorParams)]
    L.sdc_rollout_actor.argtypes = [vp, C.c_int, C.c_int, fp, fp, fp, vp, fp, fp, vp, fp, vp]
    for name in EXPORTS:
        getattr(L, name)
    built = L.sdc_version()
    if built != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} was built for ABI version {built}, this binding is written for {ABI_VERSION} "
                           "(include/sustaindc_hip.h SDC_ABI_VERSION): rebuild it with dc_rl_amd._lib.build(force=True)")
    _lib = L
    return L


class SdcError(RuntimeError):
    pass


def check(rc: int):
    if rc != 0:
        raise SdcError(load().sdc_last_error().decode())
