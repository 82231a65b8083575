#!/bin/bash
# Sub-phase costs of a wavefront's step in the COMMON-CASE two-envs-per-wavefront kernel (sdc_dynamics_fast_kernel): one build per
# pair of adjacent marks SDC_AT(i) / SDC_AT(i+1) of sdc_pairstep.hpp (only two clock stamps compiled in, so the measurement hardly
# perturbs the step; -DSDC_FAST_DEBUG=1 keeps the stamps' read-out in the common-case kernel), all run on ONE box.
#   tools/phase_scan.sh            -> table of mean / p50 / p90 / p99 microseconds per segment (4096 envs, full rings)
set -e
cd /root/repo
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
mkdir -p tools/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm -DSDC_FAST_DEBUG=1"
/opt/rocm/bin/hipcc $F -c dc_rl_amd/csrc/sdc_capi.hip -o tools/bin/capi_fd.o 2>/dev/null &
N=1
for i in $(seq 1 17); do
  j=$((i + 1))
  /opt/rocm/bin/hipcc $F -DSDC_STAMP_A=$i -DSDC_STAMP_B=$j -c dc_rl_amd/csrc/sdc_step.hip -o tools/bin/step_seg$i.o 2>/dev/null &
  N=$((N + 1))
  if [ $((N % 7)) -eq 0 ]; then wait; fi
done
wait
OTHERS=$(ls dc_rl_amd/csrc/build/*.o | grep -v -e sdc_step.o -e sdc_capi.o)
for i in $(seq 1 17); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_seg$i.so tools/bin/capi_fd.o tools/bin/step_seg$i.o $OTHERS &
done
wait
cat > tools/bin/phase_scan_run.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench
N = 4096
eng, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=8)
g = torch.Generator(device="cpu").manual_seed(1234)
pool = torch.randint(0, 3, (256, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
eng.reset()
for i in range(10300): eng.step(pool[i & 255])
assert eng.last_step_kernel() == "sdc_dynamics_fast_kernel", eng.last_step_kernel()
rows = []
for i in range(60):
    o, s, r, d, info = eng.step(pool[i & 255])
    rows.append(info[::2, 42].cpu().numpy() / 100.0)
a = np.concatenate(rows)
print("%-44s mean %.3f p50 %.3f p90 %.3f p99 %.3f" % (sys.argv[1], a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99)))
PY
NAMES="1:LS-queue-algebra 2:oldest-task-search+ages 3:policies+setpoint 4:rack-model 5:half-sums 6:HVAC+water 7:battery 8:time+obs-pool+history-slot 9:lane0-info+record-patch 10:dyn-end-to-reward-start 11:header-reads+arrivals 12:sums+outside-tests+window-updates 13:resolve+clip-bounds 14:tail-sums 15:moments+ahead/requests 16:z+rewards 17:commit"
REMOTE="cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/orig.so; for s in $NAMES; do i=\${s%%:*}; cp tools/bin/lib_seg\$i.so dc_rl_amd/csrc/libsustaindc_hip.so; python tools/bin/phase_scan_run.py \$s 2>/dev/null; done"
T=${GTIMEOUT:-1500}
exec timeout $((T + 900)) /usr/local/graft/bin/gpurun --timeout $T -- "$REMOTE"
