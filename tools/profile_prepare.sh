#!/bin/bash
# The measurement builds tools/profile_round.sh swaps in on the GPU box, made HERE (hipcc cross-compiles gfx950):
#   lib_dbg.so     -DSDC_FAST_DEBUG=1   clock stamps of the common-case (two envs per wavefront) kernel readable (wave_phases / wave_timeline / wave_tail)
#   lib_rt.so      -DSDC_RT             phase stamps of sdc_reset_kernel (reset_phases.py)
#   lib_wstamps.so -DSDC_WIDE_STAMPS    lane-0 stamps of the lane-per-env kernel's two wavefronts (dev/wide_timeline.py)
set -e
cd /root/repo
mkdir -p tools/bin
rm -f tools/bin/lib_*.so tools/bin/*.o
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm"
SRCS="dc_rl_amd/csrc/sdc_capi.hip dc_rl_amd/csrc/sdc_step.hip dc_rl_amd/csrc/sdc_rollout.hip dc_rl_amd/csrc/sdc_wide.hip dc_rl_amd/csrc/sdc_features.hip dc_rl_amd/csrc/sdc_verify.hip dc_rl_amd/csrc/sdc_reset.hip"
/opt/rocm/bin/hipcc $F -DSDC_FAST_DEBUG=1 -o tools/bin/lib_dbg.so $SRCS 2>/dev/null &
/opt/rocm/bin/hipcc $F -DSDC_RT -o tools/bin/lib_rt.so $SRCS 2>/dev/null &
/opt/rocm/bin/hipcc $F -DSDC_WIDE_STAMPS -o tools/bin/lib_wstamps.so $SRCS 2>/dev/null &
wait
ls -la tools/bin/lib_dbg.so tools/bin/lib_rt.so tools/bin/lib_wstamps.so
