#!/bin/bash
# One gpurun call that produces everything profiles/ and DESIGN.md quote for a round:  bash tools/profile_round.sh r2
# (rocprofv3 passes run from /tmp with TMPDIR=/tmp; --pmc passes -- inside bench.py -- use --kernel-trace only)
set -x
TAG=${1:-r6}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (without the PMC sub-passes, which are separate processes)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 2000 --no-pmc --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/trace.err
cd $R
python tools/trace_summary.py $O/trace --last 1500 --out $O/kernel_trace_steady_state.json 2>&1 | tail -3
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*.db" -delete
# 2. the bench line itself (with its live PMC passes and the CPU baseline); keeps the counters as profiles/pmc_latest.json
SDC_WRITE_PMC=1 timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cp profiles/pmc_latest.json $O/pmc.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench.err
timeout 600 python bench.py --mixed-racks --no-cpu-baseline --no-pmc > $O/bench_mixed_racks.json 2>> $O/bench.err
# 3. side measurements DESIGN.md quotes
timeout 600 python tools/batch_scan.py > $O/batch_scan.txt 2>&1
timeout 600 python tools/qb.py > $O/quick_rates.txt 2>&1
# (the in-kernel clock stamps of the COMMON-CASE kernels need the measurement build: tools/bin/lib_dbg.so, made by
#  tools/profile_prepare.sh; without it these tools time the general kernels)
if [ -f tools/bin/lib_dbg.so ]; then cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/prod.so; cp tools/bin/lib_dbg.so dc_rl_amd/csrc/libsustaindc_hip.so; fi
timeout 600 python tools/wave_phases.py > $O/wave_phases.txt 2>&1
timeout 600 python tools/wave_timeline.py > $O/wave_timeline.txt 2>&1
timeout 600 python tools/wave_timeline.py 2048 > $O/wave_timeline_2048.txt 2>&1     # one wavefront per SIMD: a wavefront's life without a neighbour
timeout 600 python tools/wave_tail.py > $O/wave_tail.txt 2>&1
if [ -f /tmp/prod.so ]; then cp /tmp/prod.so dc_rl_amd/csrc/libsustaindc_hip.so; fi
timeout 600 python tools/rollout_rate.py > $O/rollout_rate.txt 2>&1
SDC_GROUPS=2 SDC_N=4096 timeout 600 python tools/two_streams.py > $O/two_streams.txt 2>&1
SDC_GROUPS=2 SDC_N=8192 timeout 600 python tools/two_streams.py >> $O/two_streams.txt 2>&1
timeout 600 python tools/surface_rates.py > $O/surface_rates.txt 2>&1
timeout 600 python tools/action_mix.py > $O/action_mix.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor 2>/dev/null && /tmp/launch_floor > $O/launch_floor.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o /tmp/dispatch_rate 2>/dev/null && /tmp/dispatch_rate > $O/dispatch_rate.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate2.hip -o /tmp/dispatch_rate2 2>/dev/null && /tmp/dispatch_rate2 > $O/dispatch_rate2.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/load_latency.hip -o /tmp/load_latency 2>/dev/null && /tmp/load_latency > $O/load_latency.txt 2>&1
timeout 900 python tools/step_scan.py > $O/step_scan.txt 2>&1
bash tools/pmc_actor.sh > $O/pmc_actor.txt 2>&1
bash tools/pmc_stalls.sh > $O/pmc_stalls.txt 2>&1
bash tools/pmc_latency.sh > $O/pmc_latency.txt 2>&1
# round 4: what an unchanged single-process HARL runner gets, the one-process multi-device bench, the boundary's two kernels
timeout 600 python tools/harl_loop_rate.py 48 512 4096 > $O/harl_loop_rate.txt 2>&1
timeout 600 python bench.py --gpus 2 --single-process --devices 0,0 --steps 200 --warmup 20 > $O/bench_single_process.txt 2>&1
bash tools/reset_time.sh > $O/reset_kernels.txt 2>&1
bash tools/reset_pmc.sh > $O/reset_pmc.txt 2>&1
# (phase stamps of sdc_reset_kernel: the measurement build tools/bin/lib_rt.so = the production sources with -DSDC_RT, tools/profile_prepare.sh)
if [ -f tools/bin/lib_rt.so ]; then cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/prod.so; cp tools/bin/lib_rt.so dc_rl_amd/csrc/libsustaindc_hip.so; for n in 1024 2048 4096; do timeout 200 python tools/reset_phases.py $n; done > $O/reset_phases.txt 2>&1; cp /tmp/prod.so dc_rl_amd/csrc/libsustaindc_hip.so; fi
# round 5: the lane-per-env kernel (sdc_dynamics_wide_kernel, 12 288 envs and up): rocprofv3 time + counters at 16 384 / 32 768 envs next to the
# four-envs-per-wavefront kernel on the same rings, the two wavefronts' timeline (the -DSDC_WIDE_STAMPS build), the crossover
bash tools/dev/wide_pmc.sh 32768 > $O/wide_pmc_32768.txt 2>&1
bash tools/dev/wide_pmc.sh 16384 > $O/wide_pmc_16384.txt 2>&1
SDC_DEBUG_FLAGS=4096 bash tools/dev/wide_pmc.sh 32768 > $O/quad_pmc_32768.txt 2>&1
for f in 0 4096 2048; do echo "debug_flags $f (4096: lane-per-env kernel off, 2048: forced)"; SDC_DBG=$f timeout 300 python tools/batch_scan.py 8192 12288 16384 20480 32768 65536; done > $O/wide_crossover.txt 2>&1
if [ -f tools/bin/lib_wstamps.so ]; then cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/prod.so; cp tools/bin/lib_wstamps.so dc_rl_amd/csrc/libsustaindc_hip.so; for n in 16384 32768 65536; do timeout 200 python tools/dev/wide_timeline.py $n; done > $O/wide_timeline.txt 2>&1; cp /tmp/prod.so dc_rl_amd/csrc/libsustaindc_hip.so; fi
hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && timeout 200 /tmp/valu_rates > $O/valu_rates.txt 2>&1
# round 6: the general form of the lane-per-env kernel (configs[3] mix), what the episode boundary costs by batch size, the throughput regime,
# what the last-ending workgroup of a launch did (-DSDC_WIDE_STAMPS build), how the reward normalisation was served, the boundary's kernels
timeout 300 python tools/dev/mixed_scan.py --mixed 4096 16384 32768 65536 > $O/mixed_scan.txt 2>&1
timeout 300 python tools/dev/boundary_share.py 4096 8192 16384 32768 65536 262144 > $O/boundary_share.txt 2>&1
timeout 300 python tools/batch_scan.py 65536 98304 131072 262144 > $O/batch_scan_throughput.txt 2>&1
bash tools/dev/wide_pmc.sh 262144 > $O/wide_pmc_262144.txt 2>&1
if [ -f tools/bin/lib_wstamps.so ]; then cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/prod.so; cp tools/bin/lib_wstamps.so dc_rl_amd/csrc/libsustaindc_hip.so; for n in 16384 32768; do timeout 200 python tools/dev/wide_tail.py $n; done > $O/wide_tail.txt 2>&1; cp /tmp/prod.so dc_rl_amd/csrc/libsustaindc_hip.so; fi
for n in 16384 32768; do SDC_N=$n SDC_DBG=2 SDC_STEPS=1500 timeout 280 python tools/path_hist.py; done > $O/path_hist.txt 2>&1
bash tools/reset_time.sh > $O/reset_kernels.txt 2>&1
grep -v amdgpu.ids $O/*.txt | tail -60
cut -c1-600 $O/bench.json
ls $O
