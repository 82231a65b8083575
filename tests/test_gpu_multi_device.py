"""ONE ShareVecEnv over several devices in ONE process (dc_rl_amd/multi_device.py: one handle + one stream per device,
contiguous env ranges keyed on the GLOBAL env index).  The lease has one MI355X, so `devices=[0, 0]` / `[0, 0, 0]` put
several handles on it: the job must be, bit for bit, the unsharded batch -- over an auto-reset, NumPy and device-resident
outputs, `infos` gathered over the shards.  Reference: harl/utils/envs_tools.py:49-75, harl/envs/env_wrappers.py:222-297
(one vector env for all n_threads envs, handed to a single-process runner); SURVEY.md section 8(d) config 5."""
import numpy as np
import pytest

from dc_rl_amd import _lib as L

pytestmark = pytest.mark.gpu

ENV_ARGS = {"location": "ny", "days_per_episode": 1, "partial_obs": True, "nonoverlapping_shared_obs_space": True}


@pytest.mark.parametrize("devices,N", [([0, 0], 24), ([0, 0, 0], 50)])
def test_two_handles_on_one_gpu_equal_the_unsharded_batch_numpy(devices, N):
    from dc_rl_amd import make_train_env
    multi = make_train_env("sustaindc", seed=11, n_threads=N, env_args=dict(ENV_ARGS), devices=devices)
    whole = make_train_env("sustaindc", seed=11, n_threads=N, env_args=dict(ENV_ARGS))
    assert multi.num_envs == N and len(multi.shards) == len(devices) and multi.months == whole.months
    assert [hi - lo for lo, hi in multi.ranges] == [sh.num_envs for sh in multi.shards]
    assert all(lo % 2 == 0 for lo, _ in multi.ranges)
    assert multi.observation_space[0].shape == (26,) and multi.share_observation_space[0].shape == (29,)
    om, sm, am = multi.reset()
    ow, sw, aw = whole.reset()
    np.testing.assert_array_equal(om, ow)
    np.testing.assert_array_equal(sm, sw)
    assert am.shape == aw.shape == (N, 3, 3)
    multi.accumulate_logger_sums()
    whole.accumulate_logger_sums()
    rng = np.random.default_rng(3)
    saw_done = 0
    for t in range(110):                                    # 96-step episodes: one auto-reset inside
        a = rng.integers(0, 3, size=(N, 3, 1))
        om, sm, rm, dm, im, _ = multi.step(a)
        ow, sw, rw, dw, iw, _ = whole.step(a)
        for u, v, nm in ((om, ow, "obs"), (sm, sw, "share"), (rm, rw, "rew"), (dm, dw, "done")):
            np.testing.assert_array_equal(u, v, err_msg=f"{nm} step {t}")
        np.testing.assert_array_equal(im.rows(), iw.rows())
        e = int(rng.integers(0, N))
        assert im[e][0]["bat_SOC"] == iw[e][0]["bat_SOC"] and im[e][2]["ls_action"] == iw[e][2]["ls_action"] == int(a[e, 0, 0])
        assert im[e][1]["dc_power_ub_kW"] == iw[e][1]["dc_power_ub_kW"]
        if dm.all():
            saw_done += 1
            for e in (0, N // 2, N - 1):                    # an env of the first, a middle and the last shard
                np.testing.assert_array_equal(im[e][0]["original_obs"], iw[e][0]["original_obs"])
                np.testing.assert_array_equal(im[e][0]["original_state"], iw[e][0]["original_state"])
                assert "original_obs" not in im[e][1]
            st = multi.episode_return_sums()
            r = iw.rows()[:, [L.INFO_IDX["ep_return_ls"], L.INFO_IDX["ep_return_dc"], L.INFO_IDX["ep_return_bat"]]].astype(np.float64)
            np.testing.assert_allclose(st.sums[:3].numpy(), r.sum(0), rtol=1e-12)
            assert int(st.sums[6]) == N
    assert saw_done == 1
    (s1, n1), (s2, n2) = multi.read_logger_sums(), whole.read_logger_sums()
    assert n1 == n2 == 110
    for k in s2:
        assert s1[k] == pytest.approx(s2[k], rel=1e-12), k      # (fp64 sums over the envs, in shard order)
    multi.close()
    whole.close()


def test_device_resident_outputs_per_device():
    import torch
    from dc_rl_amd import SustainDCMultiDeviceVecEnv, SustainDCVecEnv
    N = 32
    args = dict(ENV_ARGS, month=6)
    multi = SustainDCMultiDeviceVecEnv(args, n_envs=N, seed=2, months=[6] * N, devices=[0, 0], return_torch=True)
    whole = SustainDCVecEnv(args, n_envs=N, seed=2, months=[6] * N, return_torch=True)
    om, sm, _ = multi.reset()
    ow, sw, _ = whole.reset()
    assert isinstance(om, tuple) and len(om) == 2 and torch.equal(torch.cat(om), ow) and torch.equal(torch.cat(sm), sw)
    g = torch.Generator(device="cpu").manual_seed(1)
    for t in range(100):
        a = torch.randint(0, 3, (N, 3), generator=g, dtype=torch.int32).cuda()
        parts = tuple(a[lo:hi] for lo, hi in multi.ranges)     # a policy per device hands its own device's actions over (a TUPLE of tensors)
        om, sm, rm, dm, im, _ = multi.step(parts)
        ow, sw, rw, dw, iw, _ = whole.step(a)
        assert torch.equal(torch.cat(om), ow) and torch.equal(torch.cat(sm), sw) and torch.equal(torch.cat(rm), rw)
        assert torch.equal(torch.cat(dm), dw)
        if t == 3:     # a plain LIST of per-env rows is one [N, k] batch, whatever its length (here N == 32 rows, never "2 parts")
            with pytest.raises(ValueError):
                multi.step_async((a[:3], a[3:]))           # per-device parts of the wrong sizes are refused, not reshaped
        assert im[1][3][0]["bat_SOC"] == iw[multi.ranges[1][0] + 3][0]["bat_SOC"]
    multi.close()
    whole.close()


def test_bench_single_process_two_handles_on_one_gpu():
    """`bench.py --gpus 2 --single-process --devices 0,0`: the one-process form of the multi-GPU bench (two handles, two
    streams, no torch.distributed) prints the contract's ONE line; its return statistics count both shards' episodes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0",
                        "--steps", "40", "--warmup", "8", "--envs-per-gpu", "256", "--episode-steps", "96", "--repeats", "6"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["single_process"] is True and d["metric"] == "coupled env-steps/s" and d["value"] > 0
    assert d["config"]["history_len"] == 10000 and d["config"]["faults"] == 0 and d["config"]["distinct_devices"] == 1
    assert d["timed_steps"] == 240 and d["return_stats"]["episodes"] % 512 == 0 and d["return_stats"]["episodes"] > 0
    # refuses what it was not given
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert q.returncode == 4
