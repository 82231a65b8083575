"""The N > 1 path on CPU: two gloo ranks shard the env index space and all-reduce the episode-return
statistics (the job's only collective)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dc_rl_amd.distributed import ReturnStats, init_process_group, shard_range
from dc_rl_amd.envs_tools import months_for_ranks


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = init_process_group("gloo")
    assert (r, w) == (rank, world)
    n_total = 37
    lo, hi = shard_range(n_total, rank, world)
    # every rank derives its envs' months from the GLOBAL index: the union equals the single-process job
    months = months_for_ranks(hi - lo, {}, rank_offset=lo)
    st = ReturnStats.zeros()
    rng = np.random.default_rng(1234)
    all_returns = rng.normal(size=(n_total, 3))       # same array on every rank
    st.add_episode_returns(all_returns[lo:hi])
    tot = st.all_reduce()
    mean, std, n = tot.mean_std()
    dist.barrier()
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), lo=lo, hi=hi, months=np.array(months), mean=mean, std=std, n=n,
             tmax=t.numpy(), ref_mean=all_returns.mean(0), ref_std=all_returns.std(0))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_return_stats(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert int(rs[0]["lo"]) == 0 and int(rs[0]["hi"]) == int(rs[1]["lo"]) and int(rs[1]["hi"]) == 37
    months = np.concatenate([r["months"] for r in rs])
    np.testing.assert_array_equal(months, months_for_ranks(37, {}))
    for r in rs:
        assert int(r["n"]) == 37 and float(r["tmax"][0]) == 2.0
        np.testing.assert_allclose(r["mean"], r["ref_mean"], rtol=1e-12)
        np.testing.assert_allclose(r["std"], r["ref_std"], rtol=1e-10)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(args, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):     # no launcher around it
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its two ranks itself (gloo here; RCCL on
    a GPU node) and prints ONE line with n_gpus == ranks_seen == 2.  --launch-check: launch, rendezvous and the first
    all-reduce only -- the part of bench.py that runs without a device."""
    p = _run_bench(["--gpus", "2", "--launch-check"], {"SDC_DIST_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dist_backend"] == "gloo"


def test_bench_refuses_a_world_it_was_not_asked_for():
    """--gpus and WORLD_SIZE must agree (a launcher that starts fewer ranks than the job names is an error, not a
    warning), and over RCCL --gpus N needs N visible devices: non-zero exit in both cases."""
    p = _run_bench(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and b"WORLD_SIZE is 1" in p.stderr
    if not torch.cuda.is_available():
        p = _run_bench(["--gpus", "2", "--steps", "5"], {"SDC_DIST_BACKEND": "nccl"})
        assert p.returncode != 0 and b"visible devices" in p.stderr
