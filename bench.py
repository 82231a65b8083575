#!/usr/bin/env python3
"""bench.py -- coupled env-steps/s of the MI355X-native SustainDC step (BASELINE.json metric).

Workload (BASELINE.json configs[2] at N=1; configs[4] = the same per-GPU shard on 1/2/4/8 GPUs):
4096 environment instances per GPU, `dc_config.json` (20 racks), NY-profile synthetic year traces, 7-day
episodes (672 steps) with device-side auto-reset, uniform-random {0,1,2} actions pre-generated on the
device.  Before anything is timed every env's energy-history ring is brought to its 10 000-entry steady
state by running real steps (`history_fill_steps`), so each timed step normalises against the full window.

One "step" = one sdc_step() call = one pass of the hot path over the batch: actions in -> obs, share_obs,
rewards, dones, info out, auto-reset included (SURVEY.md section 8(d) `Metric`).

Timing: the timed region is R back-to-back blocks of K = --steps steps, bracketed by barrier +
torch.cuda.synchronize() on both sides; R is chosen so that the region lasts >= 200 ms (R = 1 once K alone does),
so a short `--steps 20` run measures the same steady state as a long one instead of the first launches after
an idle GPU.  `value` = envs x K x R / wall time of the region (MAX over ranks); HIP events at the block
boundaries give the per-block times (`block_ms_median` etc.) without extra synchronisation.

Roofline (`roofline`): SURVEY.md 8(d) bounds the path by HBM (8 TB/s) and prices an env-step at B(H) = 4*H + 1540 bytes, asking
for the fraction with AND without the history term because an incremental implementation does not move the 4*H bytes.  This
step is ONE kernel that keeps the reward normalisation's order statistics incrementally, so:
  `achieved` / `peak` / `frac` (= `frac_hbm_algorithmic`) = 1540 B x envs per launch / the kernel's average launch duration
      (HIP events on the launch stream over the timed region) against 8000 GB/s -- 8(d)'s figure WITHOUT the history term;
  `frac_hbm_algorithmic_with_history_term` = the same with B(10 000) = 41 540 B: bytes the kernel does not move, may exceed 1;
  `traffic` = PMC bytes per launch (FETCH_SIZE / WRITE_SIZE with the guide's gfx950 corrections), `hbm_frac` = traffic / kernel
      time / 8 TB/s (physical), `traffic_over_alg_bytes` = traffic / (1540 x envs): re-reads and layout overhead;
  `valu_busy_frac` = SQ_ACTIVE_INST_VALU x 4 / the SIMD-cycles of the launch: what the kernel is actually bound by (wave-uniform
      fp64 physics issued once per pair / quad of envs), `issue_frac` likewise for every instruction class.
The PMC counters are collected in this very run (rank 0, N = 1) by short `rocprofv3 --pmc ... --kernel-trace` passes over
`bench.py --pmc-inner`; if rocprofv3 is not usable the numbers fall back to profiles/pmc_latest.json, which carries the hash of
the kernel sources it was measured at (`pmc_source`, `pmc_current`).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).  For N > 1 it runs one rank
per GPU over RCCL: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- when WORLD_SIZE is not set -- bench.py starts those N ranks
ITSELF (a `torch.distributed.run` re-exec on 127.0.0.1 with a free port) and relays their output, so that
`python bench.py --gpus 8` never silently measures one GPU.  It exits non-zero when fewer than N devices are visible
(RCCL needs a device per rank), when WORLD_SIZE and --gpus disagree, or when the ranks that answered the first
all-reduce are not N.  SDC_DIST_BACKEND=gloo runs the N > 1 path without RCCL (ranks may then share a device:
LOCAL_RANK is taken modulo the visible device count) -- used by the 2-ranks-on-1-GPU test; `--launch-check` runs only
the launch + rendezvous + first all-reduce and prints the skeleton line (no device needed: the CPU test of this path).
"""
from __future__ import annotations

import argparse
import contextlib
import csv
import gc
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 256 * 4            # 256 CUs x 4 SIMDs
MAX_CLOCK_GHZ = 2.4
HIST_CAP = 10000
ALG_BYTES_FIXED = 1540      # B(H) = 4*H + 1540 bytes per env-step
WIDE_MIN_ENVS = 7680        # sdc_capi.hip SDC_WIDE_MIN_ENVS: single steps of a batch this large run one lane per env (sdc_wide.hip)
STEP_KERNEL = "sdc_dynamics_fast_kernel"   # the step kernel specialised for the common case, which is what this workload is
STEP_KERNEL_PREFIX = "sdc_dynamics"            # (the general kernel sdc_dynamics_kernel serves every other case)
MIN_REGION_S = 0.2
HOSTNAME = None     # set in main(): host name + the GPU's UUID (the boxes of the pool share a container host name)


def box_stamp():
    try:
        import socket
        import torch
        u = getattr(torch.cuda.get_device_properties(0), "uuid", None)
        return f"{socket.gethostname()}/{u}" if u is not None else socket.gethostname()
    except Exception:
        return None


def alg_bytes_per_env_step(h):
    return 4 * h + ALG_BYTES_FIXED


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def csrc_hash():
    """Hash of the kernel sources: PMC numbers are only valid for the binary they were measured on."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "dc_rl_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def build_engine(n_envs, episode_steps, device, seed, dc_files=("dc_config.json",), debug_flags=0, env_index_base=0,
                 location="ny", **engine_kw):
    from dc_rl_amd import dc_config, traces
    from dc_rl_amd.engine import SdcEngine
    tb = traces.synthetic_tables(location, seed=0)
    eng = SdcEngine(n_envs, episode_steps=episode_steps, device=device, auto_reset=True, seed=seed,
                    n_dc_configs=len(dc_files), debug_flags=debug_flags, env_index_base=env_index_base, **engine_kw)
    eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
    params = [dc_config.size_datacenter(f, 1, traces.max_ambient_for_sizing(location.upper())) for f in dc_files]
    for i, p in enumerate(params):
        eng.set_dc_params(i, p)
    e = env_index_base + np.arange(n_envs)                               # GLOBAL env index (sharded jobs)
    init_day = np.array([traces.get_init_day(int(m)) for m in e % 12])   # harl/utils/envs_tools.py:58-59
    eng.assign(0, e % len(dc_files), np.maximum(0, init_day - 7), np.minimum(364, init_day + 7))
    return eng, tb, params


@contextlib.contextmanager
def no_gc():
    """A timed region without the interpreter's cyclic garbage collector: a full collection of this process's heap is a
    35-55 ms host pause (measured: it landed on the second queued sdc_rollout launch of every run without the counter
    passes and read as 23 us per step instead of 9); the host is not what these regions time."""
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        gc.enable()


def actor_weights(seed=7):
    """random weights of the reference's StochasticPolicy shape (LayerNorm(26)-64-64-3, tanh) for the closed-loop lines"""
    rngw = np.random.default_rng(seed)
    return [{"ln0_gamma": 1 + 0.1 * rngw.standard_normal(26), "ln0_beta": 0.1 * rngw.standard_normal(26),
             "w1": rngw.standard_normal((64, 26)) * 0.3, "b1": 0.1 * rngw.standard_normal(64),
             "ln1_gamma": 1 + 0.1 * rngw.standard_normal(64), "ln1_beta": 0.1 * rngw.standard_normal(64),
             "w2": rngw.standard_normal((64, 64)) * 0.2, "b2": 0.1 * rngw.standard_normal(64),
             "ln2_gamma": 1 + 0.1 * rngw.standard_normal(64), "ln2_beta": 0.1 * rngw.standard_normal(64),
             "w3": rngw.standard_normal((3, 64)) * 0.2, "b3": np.zeros(3), "activation": "tanh"} for _ in range(3)]


def secondary_rate(n_envs, episode_steps, location, device, timed_steps, month=None, loops=False, dc_files=("dc_config.json",)):
    """One secondary line: the step's rate at another batch size / episode length, measured like the headline (all
    rings filled to 10 000 by real steps, i.i.d. device-resident actions, auto-resets inside the timed region).
    loops: also sdc_rollout (48 steps per launch) and the closed loop (sdc_rollout_actor) at that batch size."""
    import torch
    eng, _, _ = build_engine(n_envs, episode_steps, device, seed=4321, location=location, dc_files=dc_files)
    if month is not None:
        from dc_rl_amd import traces
        d0 = traces.get_init_day(int(month))
        eng.assign(0, np.arange(n_envs) % len(dc_files), max(0, d0 - 7), min(364, d0 + 7))
    cdev = torch.device("cuda", device)
    POOL = 512
    g = torch.Generator(device=cdev).manual_seed(99)
    pool = torch.randint(0, 3, (POOL, n_envs, 3), dtype=torch.int32, device=cdev, generator=g)
    eng.reset()
    k = 0
    for _ in range(HIST_CAP + 64):
        eng.step(pool[k % POOL]); k += 1
    with no_gc():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(timed_steps):
            eng.step(pool[k % POOL]); k += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    faults = int((eng.info[:, 37] != 0).sum().item())
    hl = int(eng.get_state("hist_len").min())
    out = {"envs": n_envs, "kernel": eng.last_step_kernel(), "episode_steps": episode_steps, "location": location, "timed_steps": timed_steps,
           "us_per_step": round(dt / timed_steps * 1e6, 2), "value": round(n_envs * timed_steps / dt, 1),
           "unit": "env-steps/s", "history_len": hl, "faults": faults}
    if loops:
        for a_, w in enumerate(actor_weights()):
            eng.set_actor(a_, w)
        for name in ("rollout", "closed_loop"):
            eng.reset()
            for i in range(16):
                eng.step(pool[i])
            # (one untimed launch first: a rollout returns fresh [K, N, ...] tensors -- half a gigabyte at 16 384 envs -- and the first
            # call after the scan's empty_cache() pays the device allocations)
            if name == "rollout":
                eng.rollout(pool[16:16 + 48])
            else:
                eng.rollout_actor(48, sample=True)
            done_steps = 0
            with no_gc():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                while done_steps < 960:
                    kk = min(48, eng.steps_to_episode_end())
                    if name == "rollout":
                        eng.rollout(pool[16:16 + kk])
                    else:
                        eng.rollout_actor(kk, sample=True)
                    done_steps += kk
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t0
            out[name] = {"steps_per_launch": 48, "us_per_step": round(dt2 / done_steps * 1e6, 2),
                         "value": round(n_envs * done_steps / dt2, 1)}
    eng.close()
    del pool
    torch.cuda.empty_cache()
    return out


def policy_rollout_rate(n_envs, episode_steps, device, dc_files, timed_steps=960):
    """`sdc_rollout` with the built-in rule-based policies on every agent slot (do-nothing ls agent, trim-and-respond, RBCBatteryAgent:
    utils/base_agents.py, utils/trim_and_respond.py:8-38, utils/rbc_agents.py:3-47) and tou_reward for the dc agent: no action
    array, 48 env-steps per call, rings filled to 10 000 by real steps of the same policies."""
    import torch
    eng, _, _ = build_engine(n_envs, episode_steps, device, seed=4322, dc_files=dc_files, policy=(1, 3, 2), reward_method=(0, 3, 0),
                             trim_and_respond_limit=34.9)
    eng.reset()
    filled = 0
    while filled < HIST_CAP + 64:
        k = min(48, eng.steps_to_episode_end())
        eng.rollout_policy(k, want_info=False)
        filled += k
    done_steps = 0
    with no_gc():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while done_steps < timed_steps:
            k = min(48, eng.steps_to_episode_end())
            out = eng.rollout_policy(k)
            done_steps += k
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    acts = torch.stack([torch.bincount(out[5][:, :, j].reshape(-1).long(), minlength=3) for j in range(3)]).cpu().tolist()
    r = {"envs": n_envs, "kernel": eng.last_step_kernel(), "dc_configs": len(dc_files), "steps_per_call": 48,
         "us_per_step": round(dt / done_steps * 1e6, 2), "value": round(n_envs * done_steps / dt, 1), "unit": "env-steps/s",
         "policy": "ls do-nothing, dc trim-and-respond (limit 34.9), bat RBC; dc reward tou_reward",
         "actions_last_call_by_slot": acts, "history_len": int(eng.get_state("hist_len").min()),
         "faults": int((eng.info[:, 37] != 0).sum().item())}
    eng.close()
    torch.cuda.empty_cache()
    return r


def cpu_baseline(tb, params, episode_steps, budget_s=12.0):
    """The fp64 C oracle (a scalar port of the reference step) on all host cores, steady-state history
    (10 000 entries), same tables / DC config / action distribution.  Bounded sample."""
    import ctypes as C
    from oracle import pyoracle as po
    from tests import gpu_helpers as G
    from tests.parity_util import host_reset_draw
    lib = po.lib()
    cores = host_cores()
    p = G.oracle_params_from_dict(params[0])
    rng = np.random.default_rng(99)
    envs = []
    for c in range(cores):
        dr = host_reset_draw(rng, tb, 174, 188, episode_steps)
        c0 = dr["c0"]
        lo, hi = max(0, c0 - 16), c0 + episode_steps + 18
        NC = (tb["C"][lo:hi] - dr["ci_min"]) / (dr["ci_max"] - dr["ci_min"])
        NT = (dr["T"][lo:hi] - dr["t_min"]) / (dr["t_max"] - dr["t_min"])
        o = po.OracleEnv(p)
        o.begin(tb["W"][lo:hi], tb["C"][lo:hi], NC, dr["T"][lo:hi], dr["WB"][lo:hi], NT, lo, dr["day"], dr["hour"],
                episode_steps)
        o.e.hist_len = HIST_CAP
        np.ctypeslib.as_array(o.e.hist)[:] = np.clip(331 + 70 * rng.standard_normal(HIST_CAP), 150, 650)
        envs.append((o, dr["day"], dr["hour"]))
    # calibrate on one core, then size the sample to ~budget_s of wall time
    acts = rng.integers(0, 3, size=(400, 3)).astype(np.int32)
    t0 = time.perf_counter()
    lib.sdco_run_steps(C.byref(envs[0][0].e), C.byref(p), acts.ctypes.data_as(C.POINTER(C.c_int32)), 400,
                       episode_steps, envs[0][1], envs[0][2], None)
    per_step = (time.perf_counter() - t0) / 400
    n = int(max(500, min(200000, budget_s / per_step)))
    acts = rng.integers(0, 3, size=(n, 3)).astype(np.int32)
    aptr = acts.ctypes.data_as(C.POINTER(C.c_int32))

    def work(o, d, h):
        lib.sdco_run_steps(C.byref(o.e), C.byref(p), aptr, n, episode_steps, d, h, None)  # ctypes drops the GIL

    th = [threading.Thread(target=work, args=e) for e in envs]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": round(cores * n / dt, 1), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "per_core": round(n / dt, 1),
            "sample": f"{n} steps x {cores} envs (one per host thread), history ring full (10000), fp64 C oracle "
                      "(one normalize_energy per step where the reference's Python runs three)",
            "reference_python_steps_per_s_per_core": 227}   # SURVEY.md section 6 (quoted, not measured here)


# ---------------------------------------------------------------------------------------------------------------------
# PMC passes: rocprofv3 over a short inner run of this script

PMC_PASSES = (
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
    ("sq", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES",
            "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY"]),
    ("grbm", ["GRBM_GUI_ACTIVE"]),
)


def _pmc_parse(dirname, last):
    out = {}
    for fn in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        vals = {}
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                if STEP_KERNEL_PREFIX in row.get("Kernel_Name", ""):
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            tail = v[-last:]
            out[k] = sum(tail) / len(tail)
            out["_launches_" + k] = len(v)
    return out


def pmc_collect(args, timeout_s=90):
    """Counters of the step kernel from separate rocprofv3 passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE
    cannot share a pass; --pmc with --kernel-trace only).  Returns (dict, error string or None)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    res, errs = {}, []
    work = tempfile.mkdtemp(prefix="sdc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for name, ctrs in PMC_PASSES:
            d = os.path.join(work, name)
            cmd = [exe, "--pmc", *ctrs, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-inner", "--steps", "48", "--warmup", "16",
                   "--envs-per-gpu", str(args.envs_per_gpu), "--episode-steps", str(args.episode_steps)]
            if args.mixed_racks:
                cmd.append("--mixed-racks")
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   timeout=timeout_s)
            except subprocess.TimeoutExpired:
                errs.append(f"rocprofv3 pass '{name}' timed out; remaining passes skipped")
                break
            got = _pmc_parse(d, last=32)
            if not any(c in got for c in ctrs):
                errs.append(f"rocprofv3 pass '{name}' gave no counters (rc {p.returncode}): " +
                            p.stdout.decode(errors="replace")[-200:])
                continue
            res.update(got)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return (res or None), ("; ".join(errs) or None)


def pmc_summary(c):
    """Raw counters -> per-launch figures (gfx950 corrections of MI355X_MICROARCH.md section HBM)."""
    out = {"raw": {k: v for k, v in c.items() if not k.startswith("_")}}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = c["FETCH_SIZE"] * 1024 * 2      # KiB; gfx950 tallies 128-B read requests at 64 B
        write = c["WRITE_SIZE"] * 1024          # KiB; uncalibrated, as reported
        out.update(fetch_bytes_per_launch=fetch, write_bytes_per_launch=write, hbm_bytes_per_launch=fetch + write)
    w = c.get("SQ_WAVES", 0.0)
    if w:
        out["per_wave"] = {k[len("SQ_INSTS_"):].lower(): round(c[k] / w, 1) for k in c if k.startswith("SQ_INSTS_")}
        out["waves_per_launch"] = w
    if "SQ_ACTIVE_INST_VALU" in c:
        out["valu_active_simd_cycles_per_launch"] = 4.0 * c["SQ_ACTIVE_INST_VALU"]   # SQ_ACTIVE_INST_* count quad-cycles
    if "SQ_ACTIVE_INST_ANY" in c:
        out["issue_active_simd_cycles_per_launch"] = 4.0 * c["SQ_ACTIVE_INST_ANY"]  # any instruction class executing
        if c.get("SQ_WAVE_CYCLES"):
            # share of a wavefront's life with one of ITS instructions executing (the rest: parked at s_waitcnt, or waiting to issue)
            out["wave_issue_frac"] = round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    return out


def what_bounds(issue_frac, hbm_frac):
    """What the measurements say bounds a kernel: "hbm" (physical HBM traffic at >= half of the 8 TB/s peak), "issue" (the SIMDs'
    issue ports busy for >= half of the launch) or "latency" (neither: a wavefront's dependent chain + dispatch).  SURVEY 8(d)'s
    roofline for the PATH is HBM whatever this says: `roofline.roofline`, `frac`."""
    if hbm_frac is not None and hbm_frac >= 0.5:
        return "hbm"
    if issue_frac is not None and issue_frac >= 0.5:
        return "issue"
    return "latency" if (issue_frac is not None or hbm_frac is not None) else None


def scan_roofline(n_envs, episode_steps, us_per_step, args, kernel, mixed_racks=False):
    """The counter passes of `pmc_collect` at another batch size (the four-envs-per-wavefront kernel above 5632 envs):
    the same physical figures as the headline's `roofline`, against the wall-clock time per step of that batch."""
    import argparse as _ap
    a2 = _ap.Namespace(**vars(args))
    a2.envs_per_gpu, a2.episode_steps, a2.mixed_racks = n_envs, episode_steps, mixed_racks
    c, err = pmc_collect(a2, timeout_s=150)
    if c is None:
        return {"error": err}
    pm = pmc_summary(c)
    t = us_per_step * 1e-6
    simd_cycles = N_SIMD * t * MAX_CLOCK_GHZ * 1e9
    traffic = pm.get("hbm_bytes_per_launch")
    valu = pm.get("valu_active_simd_cycles_per_launch")
    alg = ALG_BYTES_FIXED * n_envs
    issue = (pm["issue_active_simd_cycles_per_launch"] / simd_cycles) if pm.get("issue_active_simd_cycles_per_launch") else None
    out = {"roofline": "hbm", "bound": what_bounds(issue, traffic / t / 1e9 / HBM_PEAK_GBPS if traffic else None),
           "kernel": kernel,      # (what sdc_last_step_kernel() reported for the timed batch)
           "achieved": round(alg / t / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(alg / t / 1e9 / HBM_PEAK_GBPS, 5), "frac_hbm_algorithmic": round(alg / t / 1e9 / HBM_PEAK_GBPS, 5),
           "valu_busy_frac": round(valu / simd_cycles, 4) if valu else None,
           "issue_frac": (round(pm["issue_active_simd_cycles_per_launch"] / simd_cycles, 4)
                          if pm.get("issue_active_simd_cycles_per_launch") else None),
           "wave_issue_frac": pm.get("wave_issue_frac"), "traffic": traffic,
           "hbm_frac": round(traffic / t / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
           "alg_bytes_per_launch": alg,
           "traffic_over_alg_bytes": round(traffic / alg, 3) if traffic else None,
           "instructions_per_wavefront": pm.get("per_wave"), "wavefronts_per_launch": pm.get("waves_per_launch"),
           "us_per_step": us_per_step, "raw": pm.get("raw")}
    if err:
        out["pmc_error"] = err[:300]
    return out


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (torch.distributed.run, one node,
    127.0.0.1, a free port), relay their stdout / stderr and exit status.  Returns the process exit code."""
    import socket
    backend = os.environ.get("SDC_DIST_BACKEND", "nccl")
    if backend == "nccl" and "--launch-check" not in sys.argv:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < n_ranks:
            print(f"bench.py: --gpus {n_ranks} needs {n_ranks} visible devices (one rank per GPU over RCCL), "
                  f"found {ndev}", file=sys.stderr)
            return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main_single_process(args):
    """`--gpus N --single-process`: N shards of 4096 envs, one C-ABI handle and one HIP stream per device, all driven by this one
    process -- every device gets its launch before any is waited for; the return statistics are added up on the host (no
    collective).  Same workload, fill, warm-up and timing rules as the one-process-per-GPU form; prints the same ONE line."""
    import torch
    D = args.gpus
    devs = [int(x) for x in args.devices.split(",")] if args.devices else list(range(D))
    if len(devs) != D:
        print(f"bench.py: --devices names {len(devs)} devices, --gpus is {D}", file=sys.stderr)
        sys.exit(4)
    nd = torch.cuda.device_count()
    if any(not 0 <= d < nd for d in devs):
        print(f"bench.py: --single-process --gpus {D} needs devices {devs}, {nd} visible", file=sys.stderr)
        sys.exit(3)
    N = args.envs_per_gpu
    dc_files = ("dc_config.json", "dc_config_r16.json", "dc_config_r25.json") if args.mixed_racks else ("dc_config.json",)
    POOL = 1024
    engs, pools, streams = [], [], []
    for r, d in enumerate(devs):
        with torch.cuda.device(d):
            st = torch.cuda.Stream(device=d)
            eng, tb, params = build_engine(N, args.episode_steps, d, seed=1234, dc_files=dc_files, env_index_base=r * N)
            eng.use_stream(st)
            g = torch.Generator(device="cpu").manual_seed(1234 + r)
            pools.append(torch.randint(0, 3, (POOL, N, 3), dtype=torch.int32, generator=g).to(torch.device("cuda", d)))
            engs.append(eng)
            streams.append(st)
    for e in engs:
        e.reset()
    ret = np.zeros(7)
    step_no, in_ep = 0, 0

    def sync():
        for st in streams:
            st.synchronize()

    def one_step():
        nonlocal step_no, in_ep
        k = step_no % POOL
        for e, p in zip(engs, pools):          # every device gets its launch ...
            e.step(p[k])
        step_no += 1
        in_ep += 1
        if in_ep == args.episode_steps:        # ... and only an episode end reads anything back (7 doubles per shard)
            in_ep = 0
            for e, st in zip(engs, streams):
                with torch.cuda.device(e.device), torch.cuda.stream(st):
                    r = e.info[:, 40:43].double()
                    ret[:] += torch.cat([r.sum(0), (r * r).sum(0), torch.tensor([float(N)], dtype=torch.float64, device=r.device)]).cpu().numpy()

    fill = 0 if args.no_fill else HIST_CAP
    for _ in range(fill + args.warmup):
        one_step()
    hlen = min(int(e.get_state("hist_len").min()) for e in engs)
    K = args.steps
    sync()
    t0 = time.perf_counter()
    for _ in range(64):
        one_step()
    sync()
    est = (time.perf_counter() - t0) / 64
    R = args.repeats if args.repeats > 0 else min(4096, max(1, int(np.ceil(MIN_REGION_S / max(1e-9, est * K)))))
    with no_gc():
        sync()
        t0 = time.perf_counter()
        for _ in range(R * K):
            one_step()
        sync()
        dt = time.perf_counter() - t0
    timed = R * K
    faults = sum(int((e.info[:, 37] != 0).sum().item()) for e in engs)
    out = {"metric": "coupled env-steps/s", "value": round(N * D * timed / dt, 1), "unit": "env-steps/s", "n_gpus": D,
           "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / timed * 1e3, 5), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64 dynamics / f32 history ring + outputs", "data": "synthetic",
           "repeats": R, "timed_steps": timed, "ranks_seen": 1, "dist_backend": None, "single_process": True,
           "config": {"workload": (f"BASELINE configs[4] form: {N * D} envs over {D} handles ({N} each) in ONE process, devices {devs}"
                                   if D > 1 else "BASELINE configs[2]: 4096 parallel envs, dc_config.json (20 racks), 1xMI355X"),
                      "envs_per_gpu": N, "episode_steps": args.episode_steps, "history_len": hlen, "history_fill_steps": fill,
                      "auto_reset": True, "actions": "i.i.d. uniform {0,1,2}, device-resident pool of 1024 steps per device",
                      "parallelism": f"env-shard x{D}, one process, one stream per device", "faults": faults,
                      "distinct_devices": len(set(devs))},
           "return_stats": {"episodes": int(ret[6]), "mean_return": [round(float(x), 3) for x in ret[0:3] / max(1.0, ret[6])],
                            "reduced": "on the host, no collective"},
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out))
    for e in engs:
        e.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--episode-steps", type=int, default=672)
    ap.add_argument("--mixed-racks", action="store_true", help="BASELINE configs[3]: 16/20/25-rack mix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes (use profiles/pmc_latest.json)")
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--no-fill", action="store_true", help="skip the history fill (debug only; invalid as a result)")
    ap.add_argument("--repeats", type=int, default=0, help="blocks of --steps in the timed region (0 = until >= 200 ms)")
    ap.add_argument("--profile-every", type=int, default=7, help="stamp the kernels' wall-clock entry / exit every k-th step (0 = off)")
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)   # the run rocprofv3 wraps
    ap.add_argument("--launch-check", action="store_true",
                    help="only launch the ranks, rendezvous, all-reduce once and print the skeleton line (no device needed)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (HARL YAML shape, batch scan, closed loop)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N devices driven from THIS process (one handle + one stream per device, no torch.distributed): "
                         "the form a single-process HARL runner uses (dc_rl_amd.multi_device)")
    ap.add_argument("--devices", type=str, default=None,
                    help="--single-process: comma-separated device ids, one per shard (default 0..N-1; '0,0' puts two handles on one GPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.single_process:
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))     # no launcher around us: start the ranks ourselves

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SDC_DIST_BACKEND", "nccl")
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: the job would not be the one asked for",
                  file=sys.stderr)
        sys.exit(4)
    if args.launch_check:
        # the launch path alone (CPU test): rendezvous, one all-reduce of ones, the skeleton of the JSON line
        seen = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo" if backend != "nccl" or not torch.cuda.is_available() else "nccl")
            t1 = torch.ones(1, dtype=torch.float64)
            if dist.get_backend() == "nccl":
                t1 = t1.cuda(local_rank)
            dist.all_reduce(t1)
            seen = int(t1.item())
        if rank == 0:
            print(json.dumps({"metric": "coupled env-steps/s", "value": None, "n_gpus": world, "ranks_seen": seen,
                              "launch_check": True, "dist_backend": (dist.get_backend() if world > 1 else None)}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(0 if seen == world else 5)
    ndev = torch.cuda.device_count()
    # the collectives' code path (rendezvous, ranks_seen, the MAX reductions, the barriers, the return-statistics all-reduce) runs
    # for every job of more than one rank -- and for ONE rank when SDC_DIST_BACKEND is set explicitly: `SDC_DIST_BACKEND=nccl
    # python bench.py --gpus 1` takes this very path through RCCL with a world of one (tests/test_gpu_distributed.py), so that
    # the first multi-GPU run meets nothing but the transport for the first time
    dist_on = world > 1 or "SDC_DIST_BACKEND" in os.environ
    if dist_on and world == 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (a documented requirement of this platform, not a tuning choice: its host driver supports dmabuf device-memory IPC only --
        # what RCCL's intra-node transports open their peers' buffers with -- and without the variable `hipIpcGetMemHandle` fails
        # with "invalid argument".  The image exports it already; this line is for a launcher that dropped it.  It cannot be
        # measured on a one-GPU lease: a world of one rank opens no peer buffer, with or without it)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and ndev < world:
            if rank == 0:
                print(f"bench.py: {world} ranks over RCCL need {world} visible devices, found {ndev}", file=sys.stderr)
            sys.exit(3)
        dev = local_rank if backend == "nccl" else local_rank % max(1, ndev)
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    else:
        dev = 0
        torch.cuda.set_device(dev)
    cdev = torch.device("cuda", dev)

    def all_reduce(t, op=None):
        """RCCL on device tensors; with gloo the (tiny) tensor takes the host round trip."""
        if not dist_on:
            return t
        op = op or dist.ReduceOp.SUM
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
        return t

    ranks_seen = 1
    if dist_on:
        ranks_seen = int(all_reduce(torch.ones(1, dtype=torch.float64, device=cdev)).item())
        if ranks_seen != world:
            if rank == 0:
                print(f"bench.py: {ranks_seen} ranks answered the all-reduce, expected {world}", file=sys.stderr)
            sys.exit(5)

    global HOSTNAME
    HOSTNAME = box_stamp()
    N = args.envs_per_gpu
    dc_files = ("dc_config.json", "dc_config_r16.json", "dc_config_r25.json") if args.mixed_racks else ("dc_config.json",)
    # one job seed; the reset RNG is keyed on the GLOBAL env index, so the job is the same set of environments
    # whatever the number of GPUs it is sharded over
    # (kernel-selection flags only for the counter passes a tool wraps -- tools/dev/wide_pmc.sh -- never for a reported line)
    dbg = int(os.environ.get("SDC_DEBUG_FLAGS", "0")) if args.pmc_inner else 0
    eng, tb, params = build_engine(N, args.episode_steps, dev, seed=1234, dc_files=dc_files, env_index_base=rank * N, debug_flags=dbg)

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)   # seed 1234
    # actions: i.i.d. uniform per env and step.  A pool of 1024 pre-generated steps indexed by a GLOBAL step counter --
    # a short cycle (say one block of --steps 20 repeated) gives every env a periodic action sequence, which narrows
    # its energy history and measurably changes the workload (more window re-centrings): 24 vs 21 us per step
    POOL = 1024
    pool = torch.randint(0, 3, (POOL, N, 3), dtype=torch.int32, generator=g).to(cdev)
    step_no = 0
    eng.reset()

    ret_stats = torch.zeros(7, dtype=torch.float64, device=cdev)  # sum r[3], sum r^2[3], episodes
    steps_in_episode = 0
    EP_COLS = slice(40, 43)
    n_collectives = 0

    def one_step(_i=None):
        nonlocal steps_in_episode, n_collectives, step_no
        obs, share, rew, done, info = eng.step(pool[step_no % POOL])
        step_no += 1
        steps_in_episode += 1
        if steps_in_episode == args.episode_steps:   # every env finished: fixed-length episodes in lock-step
            steps_in_episode = 0
            r = info[:, EP_COLS].double()
            st = torch.cat([r.sum(0), (r * r).sum(0), torch.tensor([float(N)], dtype=torch.float64, device=r.device)])
            all_reduce(st)                           # RCCL: the only collective
            n_collectives += 1
            ret_stats.add_(st)

    fill = 0 if args.no_fill else HIST_CAP
    for i in range(fill):
        one_step(i)
    for i in range(args.warmup):
        one_step(i)
    hlen = int(eng.get_state("hist_len").min())

    if args.pmc_inner:        # the short run the counter passes wrap: steady-state launches only, nothing printed
        for i in range(args.steps):
            one_step(i)
        torch.cuda.synchronize()
        eng.close()
        return

    # ---- calibrate the number of blocks: the timed region should last >= MIN_REGION_S ---------------------------------
    K = args.steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(64):
        one_step(i)
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / 64
    R = args.repeats if args.repeats > 0 else max(1, int(np.ceil(MIN_REGION_S / max(1e-9, est * K))))
    if dist_on:   # every rank times the same number of blocks
        R = int(all_reduce(torch.tensor([float(R)], dtype=torch.float64, device=cdev), dist.ReduceOp.MAX).item())
    R = min(R, 4096)

    eng.profile(args.profile_every)   # in-kernel wall-clock stamps of every k-th timed step (sdc_profile_enable)
    eng.profile_read(reset=True)
    coll0 = n_collectives
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream (the engine launches on torch's current stream) at the block boundaries
    # (an event record is a system-scope release on this runtime -- the launch after it starts with cold caches, ~70 us
    # -- so events are at least EV_STEPS steps apart: a group of blocks per event when K is small)
    EV_STEPS = 1000
    bpe = max(1, -(-EV_STEPS // K))            # blocks per event interval
    evs = [torch.cuda.Event(enable_timing=True)]
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    evs[0].record()
    for b in range(R):
        for i in range(K):
            one_step(i)
        if (b + 1) % bpe == 0 or b == R - 1:
            evs.append(torch.cuda.Event(enable_timing=True))
            evs[-1].record()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    nb = [min(bpe, R - j * bpe) for j in range(len(evs) - 1)]            # blocks in each event interval
    block_ms = np.array([evs[j].elapsed_time(evs[j + 1]) / nb[j] for j in range(len(evs) - 1)])   # per block of K steps
    ev_ms = float(sum(evs[j].elapsed_time(evs[j + 1]) for j in range(len(evs) - 1)))
    prof = eng.profile_read(reset=True)
    if prof["steps"] == 0:     # profiling was off in the timed region: sample the kernels in a short extra pass
        eng.profile(1)
        for i in range(64):
            one_step(i)
        prof = eng.profile_read(reset=True)
    eng.profile(0)
    faults = int((eng.info[:, 37] != 0).sum().item())
    # every rank's own clock, device and slowest-rank bookkeeping (so that a SCALE record explains itself): gathered as numbers
    props = torch.cuda.get_device_properties(cdev)
    mine = torch.tensor([float(rank), dt, ev_ms * 1e-3, float(props.multi_processor_count), props.total_memory / 2.0**30, float(faults),
                         float(dev)], dtype=torch.float64, device=cdev)
    per_rank = [mine.cpu()]
    dev_names = [props.name]
    if dist_on:
        if backend == "nccl":
            bufs = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(bufs, mine)
            per_rank = [b.cpu() for b in bufs]
        else:
            bufs = [torch.zeros(mine.numel(), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(bufs, mine.cpu())
            per_rank = bufs
        try:
            names = [None] * world
            dist.all_gather_object(names, f"{props.name} [{getattr(props, 'gcnArchName', '')}] on {HOSTNAME}")
            dev_names = names
        except Exception as e:      # (device names are a courtesy: the numbers above are the record)
            dev_names = [f"{props.name} (gather failed: {e!r})"] * world
        dt = float(all_reduce(torch.tensor([dt], dtype=torch.float64, device=cdev), dist.ReduceOp.MAX).item())
    else:
        dev_names = [f"{props.name} [{getattr(props, 'gcnArchName', '')}] on {HOSTNAME}"]
    fallbacks = [int((eng.info[:, 39] == v).sum().item()) for v in (1, 3, 2)]
    timed_steps = K * R

    # the amortised cost of the episode boundary (device-side reset + the episode's feature rows), measured on its own:
    # it is inside `value` (auto-resets happen in the timed region) but not inside the step kernel's duration
    reset_us = None
    if world == 1:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.reset()
        e1.record()
        torch.cuda.synchronize()
        reset_us = e0.elapsed_time(e1) / 3 * 1e3
        steps_in_episode = 0

    if rank == 0:
        total_envs = N * world
        value = total_envs * timed_steps / dt
        b = alg_bytes_per_env_step(hlen)
        nst = max(1, prof["steps"])
        k_dyn = prof["dynamics_ms"] / nst * 1e-3     # first wavefront's entry -> last one's exit, in-kernel stamps
        # the step is ONE kernel: HIP-event time over the timed launches / launches = its average launch duration
        # with the dispatch gaps between back-to-back launches (and the ~1/672 auto-resets) included
        k_evt = ev_ms * 1e-3 / timed_steps
        out = {
            "metric": "coupled env-steps/s", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(dt / timed_steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 dynamics / f32 history ring + outputs", "data": "synthetic",
            "repeats": R, "timed_steps": timed_steps, "ranks_seen": ranks_seen,
            "dist_backend": (dist.get_backend() if dist_on else None),
            # one entry per rank, from the rank's OWN clock around the same timed region (`value` uses the slowest: max over ranks)
            "per_rank": [{"rank": int(t[0]), "value": round(N * timed_steps / float(t[1]), 1), "ms_per_step": round(float(t[1]) / timed_steps * 1e3, 5),
                          "kernel_event_ms_per_step": round(float(t[2]) / timed_steps * 1e3, 5), "device": dev_names[i] if i < len(dev_names) else None,
                          "device_index": int(t[6]), "compute_units": int(t[3]), "xcds": int(t[3]) // 32, "hbm_GiB": round(float(t[4]), 1),
                          "envs": N, "faults": int(t[5])} for i, t in enumerate(per_rank)],
            "slowest_rank": int(max(per_rank, key=lambda t: float(t[1]))[0]),
            "sum_of_rank_values": round(sum(N * timed_steps / float(t[1]) for t in per_rank), 1),
            "block_ms_median": round(float(np.median(block_ms)), 5), "block_ms_min": round(float(block_ms.min()), 5),
            "block_ms_max": round(float(block_ms.max()), 5),
            "config": {"workload": ("4096 envs x mixed 16/20/25-rack dc configs" if args.mixed_racks else
                                    "BASELINE configs[2]: 4096 parallel envs, dc_config.json (20 racks), 1xMI355X")
                       if world == 1 else f"BASELINE configs[4]: {total_envs} envs sharded {world}xMI355X ({N}/GPU)",
                       "envs_per_gpu": N, "episode_steps": args.episode_steps, "history_len": hlen,
                       "history_fill_steps": fill, "auto_reset": True, "actions": "i.i.d. uniform {0,1,2}, device-resident pool of 1024 steps",
                       "parallelism": f"env-shard x{world}", "faults": faults,
                       "return_stats_all_reduces_in_timed_region": n_collectives - coll0,
                       "reward_state_last_step": {"envs_recentring_inline": fallbacks[0], "envs_rebuilding": fallbacks[1],
                                                  "envs_taking_over_a_deferred_recentred_window": fallbacks[2]}},
            "return_stats": {"episodes": int(ret_stats[6].item()),
                             "mean_return": [round(float(x), 3) for x in (ret_stats[0:3] / max(1.0, float(ret_stats[6].item())))]},
        }
        if reset_us is not None:
            out["episode_boundary"] = {"reset_plus_features_us": round(reset_us, 1),
                                       "amortised_us_per_step": round(reset_us / args.episode_steps, 3),
                                       "note": "sdc_reset_kernel + sdc_features_kernel, once per episode; inside `value`"}

        # ---- roofline of the step kernel ---------------------------------------------------------------------------------
        pmc, pmc_err, pmc_src = None, None, None
        here = csrc_hash()
        latest = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if world == 1 and not args.no_pmc:
            c, pmc_err = pmc_collect(args)
            if c is not None:
                pmc, pmc_src = pmc_summary(c), "this run (rocprofv3 --pmc passes over bench.py --pmc-inner)"
                pmc["csrc_sha"] = here
                pmc["host"] = HOSTNAME
        if pmc is None and os.path.exists(latest):
            try:
                pmc = json.load(open(latest))
                pmc_src = f"profiles/pmc_latest.json (measured at csrc_sha {pmc.get('csrc_sha')})"
            except Exception as e:
                pmc_err = (pmc_err or "") + f"; pmc_latest.json unreadable: {e!r}"
        traffic = pmc.get("hbm_bytes_per_launch") if pmc else None
        valu_cyc = pmc.get("valu_active_simd_cycles_per_launch") if pmc else None
        simd_cycles = N_SIMD * k_evt * MAX_CLOCK_GHZ * 1e9            # SIMD-cycles the launch occupies at max clock
        valu_busy = (valu_cyc / simd_cycles) if valu_cyc else None
        eff = b * N / k_evt / 1e9
        alg_fixed = ALG_BYTES_FIXED * N
        issue_frac = (pmc["issue_active_simd_cycles_per_launch"] / simd_cycles
                      if pmc and pmc.get("issue_active_simd_cycles_per_launch") else None)
        roof = {
            # `roofline`: the roofline SURVEY 8(d) prices the path against (`frac`); `bound`: what the counters say bounds THIS kernel
            "roofline": "hbm", "bound": what_bounds(issue_frac, traffic / k_evt / 1e9 / HBM_PEAK_GBPS if traffic else None) or "issue",
            "kernel": eng.last_step_kernel(),
            # algorithmic bytes per launch (WITHOUT the history term: the kernel keeps the order statistics
            # incrementally) / the kernel's average launch duration, against the 8 TB/s HBM peak
            "achieved": round(alg_fixed / k_evt / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(alg_fixed / k_evt / 1e9 / HBM_PEAK_GBPS, 5),
            "frac_hbm_algorithmic": round(alg_fixed / k_evt / 1e9 / HBM_PEAK_GBPS, 5),
            "frac_hbm_algorithmic_with_history_term": round(eff / HBM_PEAK_GBPS, 4),
            "traffic": traffic,
            "hbm_frac": round(traffic / k_evt / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
            "hbm_achieved_GBps": round(traffic / k_evt / 1e9, 1) if traffic else None,
            "traffic_over_alg_bytes": round(traffic / alg_fixed, 3) if traffic else None,
            # what the kernel is bound by: SIMD-cycles with a VALU instruction executing per launch (PMC SQ_ACTIVE_INST_VALU x 4)
            # against the SIMD-cycles of the launch: 1024 SIMDs x kernel_avg_us x 2.4 GHz
            "valu_busy_frac": round(valu_busy, 4) if valu_busy else None,
            "valu_active_M_simd_cycles_per_launch": round(valu_cyc / 1e6, 3) if valu_cyc else None,
            "available_M_simd_cycles_per_launch": round(simd_cycles / 1e6, 3),
            # every instruction class (VALU, scalar, LDS, memory, branch) against the same SIMD-cycles: what the two wavefronts
            # of a SIMD keep its issue port busy with over the whole launch, launch gap and dispatch ramp included
            "issue_frac": round(issue_frac, 4) if issue_frac else None,
            "wave_issue_frac": pmc.get("wave_issue_frac") if pmc else None,
            "kernel_avg_us": round(k_evt * 1e6, 2),
            "kernel_first_entry_to_last_exit_us": round(k_dyn * 1e6, 2),
            "timed_launches": timed_steps,
            "instructions_per_wavefront": pmc.get("per_wave") if pmc else None,
            "wavefronts_per_launch": pmc.get("waves_per_launch") if pmc else None,
            "alg_bytes_per_env_step": ALG_BYTES_FIXED, "alg_bytes_per_launch": alg_fixed,
            "alg_bytes_per_env_step_with_history_term": b,
            "pmc_source": pmc_src, "pmc_current": bool(pmc and pmc.get("csrc_sha") == here), "csrc_sha": here,
            # where the counter passes ran against where the timed region ran (boxes of the pool differ by 1-3 %: a kernel
            # figure from one box next to a wall clock from another can differ by that much in either direction)
            "timed_on_host": HOSTNAME, "pmc_on_host": (pmc.get("host") if pmc else None),
            "note": "frac = SURVEY 8(d)'s algorithmic-HBM fraction WITHOUT the history term (1540 B per env-step / kernel time / 8 TB/s); "
                    "hbm_frac = PMC bytes / kernel time / 8 TB/s (physical); traffic_over_alg_bytes = their ratio; "
                    "valu_busy_frac = what bounds this kernel (physical, <= 1); frac_hbm_algorithmic_with_history_term prices the "
                    "REFERENCE algorithm's bytes (whole history window read every step), which this kernel does not move -- not a "
                    "physical bandwidth (DESIGN.md section 4)",
        }
        if pmc_err:
            roof["pmc_error"] = pmc_err[:400]
        out["roofline"] = roof
        if pmc and pmc_src and pmc_src.startswith("this run") and os.environ.get("SDC_WRITE_PMC"):
            json.dump(pmc, open(latest, "w"), indent=1)

        if world == 1 and not args.no_rollout:
            # next to the headline (one launch per step): sdc_rollout, 48 env-steps per launch for action sequences
            # known up front (scripted policies); same work per step, every step's outputs written
            try:
                eng.reset()
                for i in range(16):
                    eng.step(pool[i])
                Kr, done_steps = 48, 0
                with no_gc():
                    torch.cuda.synchronize()
                    tr = time.perf_counter()
                    while done_steps < 1920:
                        k = min(Kr, eng.steps_to_episode_end())
                        o = (16 + done_steps) % (POOL - Kr)
                        eng.rollout(pool[o:o + k])
                        done_steps += k
                    torch.cuda.synchronize()
                    tr = time.perf_counter() - tr
                out["rollout"] = {"steps_per_launch": Kr, "value": round(N * done_steps / tr, 1), "unit": "env-steps/s",
                                  "ms_per_step": round(tr / done_steps * 1e3, 5)}
            except Exception as e:
                out["rollout"] = {"error": repr(e)}
        if world == 1 and not args.no_rollout:
            # the CLOSED loop: sdc_rollout_actor, 48 env-steps per launch with the three agents' actor networks (the
            # reference's StochasticPolicy: LayerNorm(26) -> 64 -> 64 -> 3, fp32, happo.yaml hidden_sizes [64, 64]; random
            # weights) evaluated inside the kernel between the steps -- every env-step includes three network inferences
            try:
                for a_, w_ in enumerate(actor_weights()):
                    eng.set_actor(a_, w_)
                eng.reset()
                for i in range(16):
                    eng.step(pool[i])
                Kr, done_steps = 48, 0
                with no_gc():
                    torch.cuda.synchronize()
                    tr = time.perf_counter()
                    while done_steps < 1920:
                        k = min(Kr, eng.steps_to_episode_end())
                        _, _, _, _, _, acts_cl, _ = eng.rollout_actor(k, sample=True)
                        done_steps += k
                    torch.cuda.synchronize()
                    tr = time.perf_counter() - tr
                hist_a = torch.bincount(acts_cl.reshape(-1).long(), minlength=3).cpu().tolist()
                out["closed_loop"] = {"steps_per_launch": Kr, "value": round(N * done_steps / tr, 1), "unit": "env-steps/s",
                                      "ms_per_step": round(tr / done_steps * 1e3, 5),
                                      "policy": "3 actor networks LayerNorm(26)-64-64-3 (tanh, fp32) evaluated in the kernel every "
                                                "step, actions sampled from their softmax", "actions_last_launch": hist_a}
            except Exception as e:
                out["closed_loop"] = {"error": repr(e)}
        if world == 1 and not args.no_secondary:
            # secondary lines (not `value`): the shape the reference's shipped HARL YAML trains on, and how the rate moves
            # with the number of envs per GPU -- so that the driver's record, not only profiles/, shows both
            sec = {}
            try:
                # harl/configs/envs_cfgs/sustaindc.yaml: location ca, month 6, days_per_episode 30 (2880 steps; 1.5 GB of
                # per-episode observation rows at 4096 envs); two full episodes, so two resets, in the timed region
                sec["harl_yaml_shape"] = secondary_rate(N, 30 * 96, "ca", dev, 2 * 30 * 96, month=6)
                sec["harl_yaml_shape"]["vs_headline"] = round(sec["harl_yaml_shape"]["value"] / value, 4)
            except Exception as e:
                sec["harl_yaml_shape"] = {"error": repr(e)}
            try:
                # BASELINE configs[3]: 4096 envs x 16 / 20 / 25-rack configs by env_id % 3 -- since round 4 the common-case kernel too
                # (two envs per wavefront; every env carries its own copy of its config's scalars, SdcDev::prm_env; the four-env
                # mapping needs ONE config); parity at this size: tests/test_gpu_production_sizes.py
                sec["mixed_racks"] = secondary_rate(N, args.episode_steps, "ny", dev, 2016,
                                                    dc_files=("dc_config.json", "dc_config_r16.json", "dc_config_r25.json"))
                sec["mixed_racks"]["workload"] = "BASELINE configs[3]: 4096 envs x mixed 16/20/25-rack dc configs"
                sec["mixed_racks"]["vs_headline"] = round(sec["mixed_racks"]["value"] / value, 4)
            except Exception as e:
                sec["mixed_racks"] = {"error": repr(e)}
            MIX = ("dc_config.json", "dc_config_r16.json", "dc_config_r25.json")
            try:
                # ... and the same mix at the size of configs[4] (32 768 envs on one GPU): round 6, the lane-per-env kernel's general
                # form -- every lane its own config (sdc_wide.hip GEN); parity: tests/test_gpu_production_sizes.py
                r = secondary_rate(32768, args.episode_steps, "ny", dev, 2016, dc_files=MIX)
                r["workload"] = "BASELINE configs[3] at 32 768 envs: mixed 16/20/25-rack dc configs by env_id % 3"
                if not args.no_pmc:
                    r["roofline"] = scan_roofline(32768, args.episode_steps, r["us_per_step"], args, r.get("kernel"), mixed_racks=True)
                sec["mixed_racks_32768"] = r
            except Exception as e:
                sec["mixed_racks_32768"] = {"error": repr(e)}
            try:
                # rule-based policies inside the step + tou_reward, no action array: K launches of the general form per sdc_rollout call
                sec["policy_rollout_16384"] = policy_rollout_rate(16384, args.episode_steps, dev, MIX)
            except Exception as e:
                sec["policy_rollout_16384"] = {"error": repr(e)}
            try:
                # what an UNCHANGED single-process HARL runner gets: its per-step walk over `infos` and its buffer inserts
                # restated around envs.step (tools/harl_loop_rate.py), NumPy in / out; env_side_share = the part of the
                # loop that is not the runner's own Python
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import harl_loop_rate
                sec["harl_unchanged_loop"] = [harl_loop_rate.measure(n, steps=(40 if n >= 2048 else 100), device=dev)
                                              for n in (48, 512, 4096)]
            except Exception as e:
                sec["harl_unchanged_loop"] = {"error": repr(e)}
            scan = []
            # (2 048: two envs per wavefront; 6 144: four; from 7 680: one lane per env -- sdc_wide.hip; counters at 6 144, 32 768 and 262 144)
            # (262 144 envs = four dispatch rounds of the lane-per-env kernel: the THROUGHPUT regime, where a workgroup's loads overlap
            # other workgroups' arithmetic and the kernel's rate is set by the memory system; 39 GB of state)
            for n in (2048, 6144, 8192, 16384, 32768, 65536, 262144):
                try:
                    r = secondary_rate(n, args.episode_steps, "ny", dev, 2016, loops=n in (6144, 16384))
                    r["rate_vs_4096_envs"] = round(r["value"] / value, 4)
                    if n in (6144, 32768, 262144) and not args.no_pmc:
                        r["roofline"] = scan_roofline(n, args.episode_steps, r["us_per_step"], args, r.get("kernel"))
                    scan.append(r)
                except Exception as e:
                    scan.append({"envs": n, "error": repr(e)})
            sec["batch_scan"] = scan
            out["secondary"] = sec
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(tb, params, args.episode_steps)
            except Exception as e:  # the oracle is only the timed baseline here; never hide the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    eng.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
