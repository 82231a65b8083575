import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, ctypes as C
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=0)
a = torch.randint(0, 3, (N, 3), dtype=torch.int32).to("cuda:0")
eng.reset()
for i in range(200): eng.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300): eng.step(a)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host time per eng.step() call (queue not full): %.2f us" % ((t1 - t0) / 300 * 1e6))
# raw ctypes call with prebuilt args
p = eng._out_ptrs
args = (eng._h, C.c_void_p(a.data_ptr()), p[0], p[1], p[2], p[3], p[4], p[5], eng._stream())
f = eng.lib.sdc_step
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(300): f(*args)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host time per raw sdc_step() ctypes call: %.2f us" % ((t1 - t0) / 300 * 1e6))
