import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_actor import _torch_actor, _engine
N, steps = 64, 96
nets = [_torch_actor(100 + a, "tanh") for a in range(3)]
e = _engine(N, steps)
for a in range(3):
    sd = dict(nets[a].state_dict()); sd["activation"] = "tanh"; e.set_actor(a, sd)
obs0, _ = e.reset(); obs0 = obs0.clone()
obs, share, rew, done, info, acts, logits = e.rollout_actor(2, sample=False, want_logits=True)
for a in range(3):
    with torch.no_grad():
        ref = nets[a](obs0[:, a, :].cpu()).numpy()
    got = logits[0, :, a, :].cpu().numpy()
    print("agent", a, "err per env (first 8):", np.abs(got - ref).max(-1)[:8])
    print(" ref", ref[:2], "\n got", got[:2])
    # try hypotheses: swapped envs
    sw = got.reshape(N // 2, 2, 3)[:, ::-1].reshape(N, 3)
    print(" swapped-env err", np.abs(sw - ref).max())
