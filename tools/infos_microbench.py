"""CPU-only: the logger walk + bad_masks comprehension over the C-backed infos against plain dicts (same rows), per batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dc_rl_amd import _lib as L
from dc_rl_amd.vec_env import LazyInfos
from tools.harl_loop_rate import RunnerSide
for N in (512, 4096):
    rows = np.random.rand(N, L.INFO_DIM).astype(np.float32)
    const = [{"ls_queue_max_len": 1000, "ls_unasigned_day_load_left": 0, "dc_power_lb_kW": 1.0}] * N
    done = np.zeros(N, bool)
    acts = np.zeros((N, 3), np.int64)
    plain = tuple([{**{k: float(rows[i, j]) for k, j in L.INFO_IDX.items()}, **const[i]}] * 3 for i in range(N))
    rs = RunnerSide(N, 8, 26, 29)
    def walk(inf):
        rs.per_step(inf)
        return np.array([[0.0] if "bad_transition" in info[0].keys() and info[0]["bad_transition"] == True else [1.0] for info in inf])
    def T(f, n=30):
        f(); t = time.perf_counter()
        for _ in range(n): f()
        return (time.perf_counter() - t) / n * 1e3
    keep = [None]
    def ours():
        inf = LazyInfos(rows, acts, done, const, {})
        walk(inf)
        keep[0] = inf
    def touch():
        inf = LazyInfos(rows, acts, done, const, {})
        for i in range(N): inf[i]
        keep[0] = inf
    print(N, "plain %.3f ms   ours %.3f ms   (view creation alone %.3f ms)" % (T(lambda: walk(plain)), T(ours), T(touch)))
