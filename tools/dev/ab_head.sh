#!/bin/bash
# the working tree's kernels against the last commit's on ONE box: tools/dev/ab_head.sh '<command>' [rounds]
set -e
cd /root/repo
CMD="$1"; R=${2:-2}
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm"
rm -rf /tmp/head_src && mkdir -p /tmp/head_src tools/bin && git archive HEAD dc_rl_amd/csrc include | tar -x -C /tmp/head_src
build() {   # $1 = source root, $2 = name
  local d=$1/dc_rl_amd/csrc o=/tmp/ab_obj_$2
  rm -rf $o && mkdir -p $o
  for f in $d/*.hip; do /opt/rocm/bin/hipcc $F -c $f -o $o/$(basename $f .hip).o 2>/dev/null & done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_$2.so $o/*.o
}
build /tmp/head_src head &
build /root/repo new &
wait
[ -f tools/bin/lib_head.so ] && [ -f tools/bin/lib_new.so ] || { echo "a variant did not build"; exit 1; }
REMOTE="for rep in \$(seq 1 $R); do for n in head new; do cp tools/bin/lib_\$n.so dc_rl_amd/csrc/libsustaindc_hip.so; echo \"== \$n (pass \$rep)\"; $CMD; done; done"
T=${GTIMEOUT:-900}
exec timeout $((T + 900)) /usr/local/graft/bin/gpurun --timeout $T -- "$REMOTE"
