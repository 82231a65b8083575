#!/usr/bin/env python3
"""bench.py -- coupled env-steps/s of the MI355X-native SustainDC step (BASELINE.json metric).

Workload (BASELINE.json configs[2] at N=1; configs[4] = the same per-GPU shard on 1/2/4/8 GPUs):
4096 environment instances per GPU, `dc_config.json` (20 racks), NY-profile synthetic year traces, 7-day
episodes (672 steps) with device-side auto-reset, uniform-random {0,1,2} actions pre-generated on the
device.  Before anything is timed every env's energy-history ring is brought to its 10 000-entry steady
state by running real steps (`history_fill_steps`), so each timed step normalises against the full window.

One "step" = one sdc_step() call = one pass of the hot path over the batch: actions in -> obs, share_obs,
rewards, dones, info out, auto-reset included (SURVEY.md section 8(d) `Metric`).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run, one
rank per GPU; prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HIST_CAP = 10000
ALG_BYTES_FIXED = 1540      # B(H) = 4*H + 1540 bytes per env-step


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


def build_engine(n_envs, episode_steps, device, seed, dc_files=("dc_config.json",), debug_flags=0):
    from dc_rl_amd import dc_config, traces
    from dc_rl_amd.engine import SdcEngine
    tb = traces.synthetic_tables("ny", seed=0)
    eng = SdcEngine(n_envs, episode_steps=episode_steps, device=device, auto_reset=True, seed=seed,
                    n_dc_configs=len(dc_files), debug_flags=debug_flags)
    eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
    params = [dc_config.size_datacenter(f, 1, traces.max_ambient_for_sizing("NY")) for f in dc_files]
    for i, p in enumerate(params):
        eng.set_dc_params(i, p)
    e = np.arange(n_envs)
    init_day = np.array([traces.get_init_day(int(m)) for m in e % 12])   # harl/utils/envs_tools.py:58-59
    eng.assign(0, e % len(dc_files), np.maximum(0, init_day - 7), np.minimum(364, init_day + 7))
    return eng, tb, params


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
    envs, keep = [], []
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
            "sample": f"{n} steps x {cores} envs (one per host thread), history ring full (10000), fp64 C oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--episode-steps", type=int, default=672)
    ap.add_argument("--mixed-racks", action="store_true", help="BASELINE configs[3]: 16/20/25-rack mix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fill", action="store_true", help="skip the history fill (debug only; invalid as a result)")
    ap.add_argument("--profile-every", type=int, default=7, help="stamp the kernels' wall-clock entry / exit every k-th step (0 = off)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)
    N = args.envs_per_gpu
    dc_files = ("dc_config.json", "dc_config_r16.json", "dc_config_r25.json") if args.mixed_racks else ("dc_config.json",)
    eng, tb, params = build_engine(N, args.episode_steps, dev, seed=1234 + rank, dc_files=dc_files)

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)   # seed 1234
    pool = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to(f"cuda:{dev}")
    eng.reset()

    ret_stats = torch.zeros(7, dtype=torch.float64, device=f"cuda:{dev}")  # sum r[3], sum r^2[3], episodes
    steps_in_episode = 0
    EP_COLS = slice(40, 43)

    def one_step(i):
        nonlocal steps_in_episode
        obs, share, rew, done, info = eng.step(pool[i & 63])
        steps_in_episode += 1
        if steps_in_episode == args.episode_steps:   # every env finished: fixed-length episodes in lock-step
            steps_in_episode = 0
            r = info[:, EP_COLS].double()
            st = torch.cat([r.sum(0), (r * r).sum(0), torch.tensor([float(N)], dtype=torch.float64, device=r.device)])
            if world > 1:
                dist.all_reduce(st)                  # RCCL: the only collective
            ret_stats.add_(st)

    fill = 0 if args.no_fill else HIST_CAP
    for i in range(fill):
        one_step(i)
    for i in range(args.warmup):
        one_step(i)
    hlen = int(eng.get_state("hist_len").min())

    eng.profile(args.profile_every)   # in-kernel wall-clock stamps of every k-th timed step (sdc_profile_enable)
    eng.profile_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the launch stream (the engine launches on torch's current stream) bracket the timed launches
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        one_step(i)
    ev1.record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    prof = eng.profile_read(reset=True)
    if prof["steps"] == 0:     # profiling was off in the timed region: sample the kernels in a short extra pass
        eng.profile(1)
        for i in range(64):
            one_step(i)
        prof = eng.profile_read(reset=True)
        eng.profile(0)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    faults = int((eng.info[:, 37] != 0).sum().item())
    fallbacks = [int((eng.info[:, 39] == v).sum().item()) for v in (1, 3)]

    if rank == 0:
        total_envs = N * world
        value = total_envs * args.steps / dt
        b = alg_bytes_per_env_step(hlen)
        nst = max(1, prof["steps"])
        k_dyn = prof["dynamics_ms"] / nst * 1e-3     # average launch duration, in-kernel wall-clock stamps
        k_rst = prof["reset_ms"] / max(1, prof["resets"]) * 1e-3
        # the step is ONE kernel: its average launch duration = HIP-event time over the timed launches / launches
        # (dispatch gaps between back-to-back launches included, so this is the conservative figure; the span from
        # the first wavefront's entry to the last one's exit, from in-kernel clock stamps, is reported next to it)
        k_evt = ev_ms * 1e-3 / args.steps
        achieved = b * N / k_evt / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "coupled env-steps/s", "value": round(value, 1), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 dynamics / f32 history ring + outputs", "data": "synthetic",
            "config": {"workload": ("4096 envs x mixed 16/20/25-rack dc configs" if args.mixed_racks else
                                    "BASELINE configs[2]: 4096 parallel envs, dc_config.json (20 racks), 1xMI355X")
                       if world == 1 else f"BASELINE configs[4]: {total_envs} envs sharded {world}xMI355X (4096/GPU)",
                       "envs_per_gpu": N, "episode_steps": args.episode_steps, "history_len": hlen,
                       "history_fill_steps": fill, "auto_reset": True, "actions": "uniform {0,1,2}, device-resident",
                       "parallelism": f"env-shard x{world}", "faults": faults,
                       "ring_read_envs_last_step": {"window_recentred_ahead_of_need": fallbacks[0], "rebuild": fallbacks[1]}},
            # the one kernel of a step.  `achieved` prices the reference algorithm's bytes (the whole
            # history window is read every step); the trackers make most steps skip that read, so the HBM bytes
            # actually moved (`traffic`, PMC) are far below it and `frac` is an effective, not a physical, bandwidth.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "sdc_dynamics_kernel", "kernel_avg_us": round(k_evt * 1e6, 2),
                         "kernel_first_entry_to_last_exit_us": round(k_dyn * 1e6, 2),
                         "alg_bytes_per_env_step": b, "alg_bytes_per_launch": b * N,
                         "timed_launches": prof["steps"],
                         "other_kernels": {"sdc_reset_kernel_avg_us": round(k_rst * 1e6, 2), "auto_resets": prof["resets"]},
                         "frac_without_history_term": round(ALG_BYTES_FIXED * N / k_evt / 1e9 / HBM_PEAK_GBPS, 5),
                         "note": "effective bandwidth: algorithmic bytes of the reference's per-step history pass / "
                                 "kernel time; the kernel keeps that state incrementally (traffic = bytes really "
                                 "moved) and is fp64-VALU / latency bound (DESIGN.md section 4)"},
            "return_stats": {"episodes": int(ret_stats[6].item()),
                             "mean_return": [round(float(x), 3) for x in (ret_stats[0:3] / max(1.0, float(ret_stats[6].item())))]},
        }
        if world == 1:
            # next to the headline (one launch per step): sdc_rollout, 48 env-steps per launch for action sequences
            # known up front (scripted policies); same work per step, every step's outputs written
            try:
                K, done_steps = 48, 0
                seq = pool[:K].contiguous()
                torch.cuda.synchronize()
                tr = time.perf_counter()
                while done_steps < 960:
                    k = min(K, eng.steps_to_episode_end())
                    eng.rollout(seq[:k])
                    done_steps += k
                torch.cuda.synchronize()
                tr = time.perf_counter() - tr
                out["rollout"] = {"steps_per_launch": K, "value": round(N * done_steps / tr, 1), "unit": "env-steps/s",
                                  "ms_per_step": round(tr / done_steps * 1e3, 5)}
            except Exception as e:
                out["rollout"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(tb, params, args.episode_steps)
                out["cpu_baseline"]["reference_python_steps_per_s_per_core"] = 227
            except Exception as e:  # the oracle is only the timed baseline here; never hide the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
