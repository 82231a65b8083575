#!/bin/bash
# variants of sdc_wide.hip on ONE box: tools/dev/ab_wide.sh '<command>' name1:'-DX=1' name2:'' ...   (the other objects: the current build)
set -e
cd /root/repo
CMD="$1"; shift
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
mkdir -p tools/bin
NAMES=""
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"; rm -f tools/bin/lib_$name.so
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm $flags -c dc_rl_amd/csrc/sdc_wide.hip -o tools/bin/wide_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_$name.so $(ls dc_rl_amd/csrc/build/*.o | grep -v sdc_wide.o) tools/bin/wide_$name.o ) &
  NAMES="$NAMES $name"
done
wait
for n in $NAMES; do [ -f tools/bin/lib_$n.so ] || { echo "variant $n did not compile"; exit 1; }; done
REMOTE="cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/orig.so; for n in $NAMES; do cp tools/bin/lib_\$n.so dc_rl_amd/csrc/libsustaindc_hip.so; echo \"== \$n\"; $CMD; done"
T=${GTIMEOUT:-900}
exec timeout $((T + 900)) /usr/local/graft/bin/gpurun --timeout $T -- "$REMOTE"
