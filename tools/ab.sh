#!/bin/bash
# A/B on ONE box (boxes of the pool differ by ~1-2 %):  tools/ab.sh '<command>' name1:'-DX=1' name2:'-DX=2' ...
# builds every variant locally (hipcc, gfx950), ships them under tools/bin/, and runs the command once per variant
set -e
cd /root/repo
CMD="$1"; shift
# the production build's scheduling strategy (dc_rl_amd/_lib.py HIPCC_FLAGS); AB_BASE="" to compare strategies
BASE="${AB_BASE--mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm}"
mkdir -p tools/bin
SRCS="dc_rl_amd/csrc/sdc_capi.hip dc_rl_amd/csrc/sdc_step.hip dc_rl_amd/csrc/sdc_rollout.hip dc_rl_amd/csrc/sdc_wide.hip dc_rl_amd/csrc/sdc_features.hip dc_rl_amd/csrc/sdc_verify.hip dc_rl_amd/csrc/sdc_reset.hip"
NAMES=""
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  rm -f tools/bin/lib_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $BASE $flags -o tools/bin/lib_$name.so $SRCS 2>/dev/null &
  NAMES="$NAMES $name"
done
wait
for n in $NAMES; do [ -f tools/bin/lib_$n.so ] || { echo "variant $n did not compile"; exit 1; }; done
REMOTE="cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/orig.so; for rep in 1 2; do for n in $NAMES; do cp tools/bin/lib_\$n.so dc_rl_amd/csrc/libsustaindc_hip.so; echo \"== \$n (pass \$rep)\"; $CMD; done; done"
T=${GTIMEOUT:-900}
exec timeout $((T + 900)) gpurun --timeout $T -- "$REMOTE"
