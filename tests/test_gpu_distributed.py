"""The N > 1 path on hardware inside a one-GPU lease: two torch.distributed ranks of bench.py share
the one MI355X (gloo for the 7-double return-statistics all-reduce and the MAX timing reduction -- RCCL needs one
device per rank), and a shard of a job is the same set of environments as the same index range of the unsharded job."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(world, extra_env=None, self_launch=False):
    common = ["--steps", "40", "--warmup", "8", "--envs-per-gpu", "256", "--episode-steps", "96", "--repeats", "6",
              "--no-cpu-baseline", "--no-pmc", "--no-rollout"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    common += ["--no-secondary"]
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if world == 1 or self_launch:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + common    # no wrapper: bench.py starts its ranks
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
               "--gpus", str(world)] + common
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines          # ONE JSON line, printed by rank 0
    return json.loads(lines[0])


def test_bench_line_of_eight_ranks_explains_itself():
    """The SCALE record's line for N > 1: eight gloo ranks on the lease's one MI355X -- per-rank value / ms_per_step
    from every rank's own clock, device name / compute units / XCD count, the slowest rank; `value` is the whole job at the SLOWEST
    rank's time, so it cannot exceed the sum of the ranks' own values."""
    d = _bench(8, {"SDC_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and len(d["per_rank"]) == 8
    assert sorted(r["rank"] for r in d["per_rank"]) == list(range(8))
    for r in d["per_rank"]:
        assert set(r) >= {"rank", "value", "ms_per_step", "device", "device_index", "compute_units", "xcds", "hbm_GiB", "envs", "faults"}
        assert r["value"] > 0 and r["ms_per_step"] > 0 and r["envs"] == 256 and r["faults"] == 0
        assert r["compute_units"] == 256 and r["xcds"] == 8 and "gfx950" in r["device"]
    slow = max(d["per_rank"], key=lambda r: r["ms_per_step"])
    assert d["slowest_rank"] == slow["rank"]
    assert abs(d["ms_per_step"] - slow["ms_per_step"]) <= 1e-4 * slow["ms_per_step"] + 1e-5
    assert d["value"] <= d["sum_of_rank_values"] * (1 + 1e-6)
    assert d["config"]["parallelism"] == "env-shard x8" and d["scaling"] == "weak"


def test_bench_two_ranks_on_one_gpu_gloo():
    one = _bench(1)
    two = _bench(2, {"SDC_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["dist_backend"] == "gloo"
    assert two["scaling"] == "weak" and two["config"]["envs_per_gpu"] == 256
    assert two["config"]["parallelism"] == "env-shard x2"
    # same number of steps on every rank: the all-reduced episode count doubles, and so does the number of envs behind `value`
    assert one["timed_steps"] == two["timed_steps"] == 240
    assert two["return_stats"]["episodes"] == 2 * one["return_stats"]["episodes"] > 0
    assert two["config"]["return_stats_all_reduces_in_timed_region"] == one["config"]["return_stats_all_reduces_in_timed_region"] >= 2
    assert two["value"] > 0 and two["ms_per_step"] > 0 and two["config"]["faults"] == 0
    # rank 0 of the 2-rank job steps the same envs (global indices 0..255, same job seed) with the same actions as the
    # 1-rank job: its share of the mean return is the same, the other rank's differs but is of the same size
    m1, m2 = np.array(one["return_stats"]["mean_return"]), np.array(two["return_stats"]["mean_return"])
    assert np.all(np.abs(m2 - m1) < 0.25 * np.abs(m1) + 1.0)


def test_bench_self_launch_equals_the_launcher_form():
    """`python bench.py --gpus 2` with no wrapper (bench.py re-execs itself under torch.distributed.run) gives the job the
    driver's `torch.distributed.run ... bench.py --gpus 2` form gives: two ranks, the same envs and actions."""
    a = _bench(2, {"SDC_DIST_BACKEND": "gloo"}, self_launch=True)
    b = _bench(2, {"SDC_DIST_BACKEND": "gloo"}, self_launch=False)
    assert a["n_gpus"] == b["n_gpus"] == 2 and a["ranks_seen"] == b["ranks_seen"] == 2
    assert a["timed_steps"] == b["timed_steps"] and a["return_stats"] == b["return_stats"]
    # over RCCL a one-GPU lease cannot hold two ranks: refused, not run on one device
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SDC_DIST_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "5"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b"visible devices" in p.stderr


def _sharded_worker(rank, world, port, out_dir, n_total=24, n_steps=100):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from dc_rl_amd.distributed import ReturnStats, init_process_group, make_sharded_train_env
    init_process_group("gloo")          # (both ranks share the lease's one MI355X: LOCAL_RANK 0)
    env, (lo, hi) = make_sharded_train_env("sustaindc", 9, n_total, {"days_per_episode": 1}, return_torch=True)
    obs, share, avail = env.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    acts = torch.randint(0, 3, (n_steps, n_total, 3), dtype=torch.int32, generator=g)
    st = ReturnStats.zeros()
    outs = []
    for t in range(n_steps):             # crosses one auto-reset (96-step episodes)
        o, s, r, d, infos, av = env.step(acts[t, lo:hi].cuda())
        outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy()))
        if bool(d.any()):
            st.add_episode_returns(env.engine.info[:, 40:43].double())
    tot = st.all_reduce()
    np.savez(os.path.join(out_dir, f"s{rank}.npz"), lo=lo, hi=hi, obs=np.stack([x[0] for x in outs]),
             rew=np.stack([x[1] for x in outs]), done=np.stack([x[2] for x in outs]), months=np.array(env.months),
             episodes=int(tot.sums[6].item()))
    env.close()
    dist.destroy_process_group()


def test_make_sharded_train_env_two_ranks(tmp_path):
    """dc_rl_amd.distributed.make_sharded_train_env (the HARL-side entry of the N > 1 path: harl/utils/envs_tools.py:49-75
    with the env range sharded): two gloo ranks on the one MI355X hold envs [0, 12) and [12, 24) of a 24-env job; together
    they are the unsharded `make_train_env(..., 24)` batch -- months, observations, rewards, dones over an auto-reset --
    and the all-reduced episode count is the job's."""
    import torch
    import torch.multiprocessing as mp
    from dc_rl_amd.envs_tools import make_train_env
    mp.spawn(_sharded_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    rs = [np.load(tmp_path / f"s{r}.npz") for r in range(2)]
    assert (int(rs[0]["lo"]), int(rs[0]["hi"]), int(rs[1]["lo"]), int(rs[1]["hi"])) == (0, 12, 12, 24)
    whole = make_train_env("sustaindc", 9, 24, {"days_per_episode": 1}, return_torch=True)
    np.testing.assert_array_equal(np.concatenate([r["months"] for r in rs]), np.array(whole.months))
    whole.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    acts = torch.randint(0, 3, (100, 24, 3), dtype=torch.int32, generator=g)
    for t in range(100):
        o, s, r, d, infos, av = whole.step(acts[t].cuda())
        np.testing.assert_array_equal(o.cpu().numpy(), np.concatenate([x["obs"][t] for x in rs]), err_msg=str(t))
        np.testing.assert_array_equal(r.cpu().numpy(), np.concatenate([x["rew"][t] for x in rs]), err_msg=str(t))
        np.testing.assert_array_equal(d.cpu().numpy(), np.concatenate([x["done"][t] for x in rs]), err_msg=str(t))
    assert int(rs[0]["episodes"]) == int(rs[1]["episodes"]) == 24
    whole.close()


def test_shard_is_the_same_environments_as_the_unsharded_job():
    """Months follow the global env index (harl/utils/envs_tools.py:56-62) and the reset RNG is keyed on
    (seed, env_index_base + env, episode): shard [lo, hi) of a job draws what envs lo..hi-1 of the whole job draw."""
    import torch
    from dc_rl_amd import dc_config, traces
    from dc_rl_amd.engine import SdcEngine
    from dc_rl_amd.envs_tools import months_for_ranks
    from dc_rl_amd.distributed import shard_range
    n_total, steps, seed = 96, 48, 77
    tb = traces.synthetic_tables("ny", 0)
    p = dc_config.size_datacenter("dc_config.json", 1, 30.0)

    def mk(lo, hi):
        e = SdcEngine(hi - lo, episode_steps=steps, auto_reset=True, seed=seed, env_index_base=lo)
        e.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"])
        e.set_dc_params(0, p)
        d0 = np.array([traces.get_init_day(m) for m in months_for_ranks(hi - lo, {}, rank_offset=lo)])
        e.assign(0, 0, np.maximum(0, d0 - 7), np.minimum(364, d0 + 7))
        return e

    whole = mk(0, n_total)
    lo, hi = shard_range(n_total, 1, 2)
    shard = mk(lo, hi)
    ow, _ = whole.reset()
    os_, _ = shard.reset()
    assert torch.equal(ow[lo:hi], os_)
    g = torch.Generator(device="cpu").manual_seed(3)
    acts = torch.randint(0, 3, (60, n_total, 3), dtype=torch.int32, generator=g).cuda()
    for t in range(60):     # crosses an auto-reset (episode 2 draws)
        xw = whole.step(acts[t])
        xs = shard.step(acts[t, lo:hi].contiguous())
        for u, v in zip(xw, xs):
            assert torch.equal(u[lo:hi], v), t
    for name in ("day", "hourq", "cursor", "t_min", "t_den", "episode"):
        np.testing.assert_array_equal(whole.get_state(name)[lo:hi], shard.get_state(name))
    np.testing.assert_array_equal(whole.get_state("t_win")[lo:hi], shard.get_state("t_win"))
    assert (whole.get_state("episode") == 2).all()
    whole.close()
    shard.close()


_RCCL_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from dc_rl_amd.distributed import ReturnStats, init_process_group
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r},
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
# the path's one collective (SURVEY.md section 8(e)): the 7-double return-statistics all-reduce, on RCCL
st = ReturnStats.zeros(dev)
st.add_episode_returns(torch.tensor([[1.0, 2.0, 3.0], [3.0, 2.0, 1.0]]))
tot = st.all_reduce()
mean, std, n = tot.mean_std()
# ... the host-resident form (the multi-device env adds its shards up on the host) through the same backend
host = ReturnStats.zeros()
host.add_episode_returns([[4.0, 4.0, 4.0]])
hn = int(host.all_reduce().sums[6])
# ... and what bench.py reduces around its timed region (MAX over ranks)
tmax = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print(json.dumps(dict(backend=dist.get_backend(), n=n, mean=mean.tolist(), std=std.tolist(), host_n=hn, tmax=float(tmax.item()),
                      rccl=str(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None)))
dist.destroy_process_group()
"""


def test_rccl_world_size_1():
    """RCCL itself, on the leased GPU: backend "nccl" with a world of ONE rank -- library load, communicator init, the path's
    7-double all-reduce through ReturnStats.all_reduce, a MAX reduction and a barrier.  (A world of N needs a device per rank:
    that is the driver's 8-GPU run; everything up to the transport is exercised here.)"""
    p = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT.format(root=ROOT, port=str(_free_port()))], cwd=ROOT,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    out = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    print("RCCL world of one:", out)
    assert out["backend"] == "nccl" and out["n"] == 2 and out["host_n"] == 1 and out["tmax"] == 1.25
    np.testing.assert_allclose(out["mean"], [2, 2, 2])
    np.testing.assert_allclose(out["std"], [1, 0, 1])



def test_eight_gloo_ranks_on_one_gpu_equal_the_unsharded_4096_env_batch(tmp_path):
    """BASELINE configs[4]'s shape at one eighth the size per rank, on the lease's one MI355X: EIGHT torch.distributed ranks (gloo:
    RCCL needs a device per rank) hold 512 envs each of a 4096-env job through `make_sharded_train_env`; concatenated they are the
    unsharded `make_train_env(..., 4096)` batch bit for bit -- months, observations, rewards, dones over an auto-reset -- and every
    rank's all-reduced episode count is the job's.  What this leaves untested without peers: the xGMI transport under RCCL only."""
    import torch
    import torch.multiprocessing as mp
    from dc_rl_amd.envs_tools import make_train_env
    W, NT, STEPS = 8, 4096, 100
    mp.spawn(_sharded_worker, args=(W, _free_port(), str(tmp_path), NT, STEPS), nprocs=W, join=True)
    rs = [np.load(tmp_path / f"s{r}.npz") for r in range(W)]
    assert [(int(r["lo"]), int(r["hi"])) for r in rs] == [(512 * i, 512 * (i + 1)) for i in range(W)]
    whole = make_train_env("sustaindc", 9, NT, {"days_per_episode": 1}, return_torch=True)
    np.testing.assert_array_equal(np.concatenate([r["months"] for r in rs]), np.array(whole.months))
    whole.reset()
    g = torch.Generator(device="cpu").manual_seed(4)
    acts = torch.randint(0, 3, (STEPS, NT, 3), dtype=torch.int32, generator=g)
    for t in range(STEPS):
        o, s, r, d, infos, av = whole.step(acts[t].cuda())
        np.testing.assert_array_equal(o.cpu().numpy(), np.concatenate([x["obs"][t] for x in rs]), err_msg=str(t))
        np.testing.assert_array_equal(r.cpu().numpy(), np.concatenate([x["rew"][t] for x in rs]), err_msg=str(t))
        np.testing.assert_array_equal(d.cpu().numpy(), np.concatenate([x["done"][t] for x in rs]), err_msg=str(t))
    assert [int(x["episodes"]) for x in rs] == [NT] * W
    whole.close()


def test_bench_one_rank_through_rccl():
    """`SDC_DIST_BACKEND=nccl python bench.py --gpus 1`: bench.py's OWN collective path -- rendezvous with `device_id=`, the
    ranks_seen all-reduce, the MAX reductions of the block count and of `dt`, the barriers either side of the timed region, the
    return-statistics all-reduce at every episode end, destroy -- through RCCL with a world of one rank.  Same envs, same actions,
    same statistics as the plain one-rank run."""
    plain = _bench(1)
    rccl = _bench(1, {"SDC_DIST_BACKEND": "nccl"})
    assert plain["dist_backend"] is None and rccl["dist_backend"] == "nccl"
    assert rccl["n_gpus"] == 1 and rccl["ranks_seen"] == 1 and rccl["value"] > 0 and rccl["config"]["faults"] == 0
    assert rccl["timed_steps"] == plain["timed_steps"] == 240
    assert rccl["config"]["return_stats_all_reduces_in_timed_region"] >= 2
    assert rccl["return_stats"] == plain["return_stats"]


def test_bench_eight_shards_in_one_process_on_one_gpu():
    """`bench.py --gpus 8 --single-process --devices 0,0,0,0,0,0,0,0`: the one-process form of the 8-GPU job (eight C-ABI handles,
    eight streams) with every shard on the lease's one device."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--single-process", "--devices",
                        ",".join(["0"] * 8), "--steps", "40", "--warmup", "8", "--envs-per-gpu", "512", "--episode-steps", "96",
                        "--repeats", "3"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["single_process"] is True and d["value"] > 0 and d["config"]["faults"] == 0
    assert d["config"]["distinct_devices"] == 1 and d["timed_steps"] == 120
    assert d["return_stats"]["episodes"] % 4096 == 0 and d["return_stats"]["episodes"] > 0
