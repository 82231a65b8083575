"""The load-shifting queue's oldest-task search on scripted action patterns (envs/carbon_ls.py:172-324): the step finds
the new oldest task among the 32 queue-table entries from the old one on (requested with the step's inputs when the
action can pop tasks) and falls back to a 32-ary search over the table when it moved further -- long idle gaps between
bursts of deferred tasks, overdue tasks popped under an action that is not "process", queues that run empty and refill.
Step by step against the fp64 CPU oracle (1e-5, verify mode on); the info block carries the queue statistics."""
import numpy as np
import pytest

from tests import gpu_helpers as G
from tests import parity_util as P

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _ls_script(env, t):
    """defer bursts separated by idle gaps of different lengths per env, then long processing runs"""
    period = 40 + 23 * env                # 40 .. 339 steps
    burst = 4 + (env % 5)                 # steps of deferring at the start of each period
    gap = min(period - burst - 8, 12 + 9 * env)   # idle steps after the burst (up to > 97: tasks become overdue)
    ph = t % period
    if ph < burst:
        return 0
    if ph < burst + gap:
        return 1
    return 2


def test_oldest_task_search_gaps_overdue_and_empty_queue_vs_oracle():
    N, steps = 14, 672
    rig = P.ParityRig(N, episode_steps=steps, seed=97)
    arng = np.random.default_rng(98)
    worst = dict(obs=0.0, rew=0.0, info=0.0)
    eobs, oobs = rig.reset_all()
    for i, o in oobs.items():
        worst["obs"] = max(worst["obs"], float(G.rel_err(eobs[i], o).max()))
    seen_overdue = seen_empty = seen_far = 0
    prev_oldest = np.zeros(N)
    for t in range(steps):
        acts = arng.integers(0, 3, size=(N, 3)).astype(np.int32)
        acts[:, 0] = [_ls_script(i, t) for i in range(N)]
        if t % 7 == 3:
            acts[N - 1, 0] = 2            # one env mixes in extra processing steps
        P.compare_step(rig, acts, worst)
        info = rig.eng.info.cpu().numpy()
        from dc_rl_amd import _lib as L
        od = info[:, L.INFO_IDX["ls_overdue_penalty"]]
        q = info[:, L.INFO_IDX["ls_tasks_in_queue"]]
        oldest = info[:, L.INFO_IDX["ls_oldest_task_age"]] * 24.0 * 4.0     # steps
        seen_overdue += int((od > 0).sum())
        seen_empty += int((q == 0).sum())
        seen_far += int(((prev_oldest - oldest) > 32).sum())      # the oldest task moved by more than the 32 prefetched entries
        prev_oldest = oldest
    print("queue paths:", worst, "overdue env-steps", seen_overdue, "empty", seen_empty, "jumps > 32 steps", seen_far)
    assert seen_overdue > 20 and seen_empty > 20 and seen_far > 5
    assert worst["obs"] <= TOL and worst["rew"] <= TOL and worst["info"] <= 2e-6
    assert (rig.eng.get_state("order_stat_sticky") == 0).all()
    rig.eng.close()
