#!/bin/bash
# build variants of the library from -D flags applied to the listed translation units (the other objects: the current build):
#   tools/dev/ab_build.sh "sdc_wide sdc_capi" name1:'-DX=1' name2:'-DX=2' ...   -> tools/bin/lib_<name>.so   (then tools/dev/ab_libs.sh)
cd /root/repo
TUS="$1"; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm"
mkdir -p tools/bin
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"; rm -f tools/bin/lib_$name.so
  (
    objs=""; skip=""
    for tu in $TUS; do
      /opt/rocm/bin/hipcc $FL $flags -c dc_rl_amd/csrc/$tu.hip -o tools/bin/${tu}_$name.o 2>/dev/null || exit 1
      objs="$objs tools/bin/${tu}_$name.o"; skip="$skip -e /$tu.o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/lib_$name.so $(ls dc_rl_amd/csrc/build/*.o | grep -v $skip) $objs && echo built $name
  ) &
done
wait
