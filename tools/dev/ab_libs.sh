#!/bin/bash
# run a command once per ALREADY BUILT library tools/bin/lib_<name>.so on one box, two passes:  tools/dev/ab_libs.sh '<command>' name1 name2 ...
cd /root/repo
CMD="$1"; shift
NAMES="$@"
for n in $NAMES; do [ -f tools/bin/lib_$n.so ] || { echo "no tools/bin/lib_$n.so"; exit 1; }; done
REMOTE="cp dc_rl_amd/csrc/libsustaindc_hip.so /tmp/orig.so; for rep in 1 2; do for n in $NAMES; do cp tools/bin/lib_\$n.so dc_rl_amd/csrc/libsustaindc_hip.so; echo \"== \$n\"; $CMD; done; done; cp /tmp/orig.so dc_rl_amd/csrc/libsustaindc_hip.so"
T=${GTIMEOUT:-900}
exec timeout $((T + 900)) /usr/local/graft/bin/gpurun --timeout $T -- "$REMOTE"
