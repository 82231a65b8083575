import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
# (1) PCIe-inclusive rate through the HARL surface with NumPy outputs
from dc_rl_amd.envs_tools import make_train_env
env_args = {"location": "ny", "month": 6, "days_per_episode": 7}
envs = make_train_env("sustaindc", 1, 4096, env_args)
obs, share, avail = envs.reset()
acts = np.random.default_rng(0).integers(0, 3, size=(64, 4096, 3)).astype(np.int32)
for i in range(50): envs.step(acts[i & 63])
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 300
for i in range(K): o, s, r, d, info, a = envs.step(acts[i & 63])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("HARL surface, NumPy outputs (device->host copies every step): %.1f us/step, %.1f M env-steps/s" % (dt / K * 1e6, 4096 * K / dt / 1e6))
envs.close()
# (1b) the same surface with device-resident outputs and actions (a GPU policy): no copies, no host synchronisation
envs = make_train_env("sustaindc", 1, 4096, env_args, return_torch=True)
envs.reset()
acts_t = torch.from_numpy(acts).cuda()
for i in range(50): envs.step(acts_t[i & 63])
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 2000
for i in range(K): o, s, r, d, info, a = envs.step(acts_t[i & 63])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("HARL surface, torch outputs (everything stays on the device): %.1f us/step, %.1f M env-steps/s" % (dt / K * 1e6, 4096 * K / dt / 1e6))
envs.close()
# (2) single env latency through SustainDC
from dc_rl_amd import SustainDC
e = SustainDC({"location": "ny", "month": 6, "days_per_episode": 7})
e.reset()
for i in range(50): e.step({"agent_ls": 1, "agent_dc": 1, "agent_bat": 2})
t0 = time.perf_counter()
for i in range(300): e.step({"agent_ls": 1, "agent_dc": 1, "agent_bat": 2})
dt = time.perf_counter() - t0
print("SustainDC (1 env, dict API, host sync every step): %.1f us/step = %.0f steps/s" % (dt / 300 * 1e6, 300 / dt))
e.close()
