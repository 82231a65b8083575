#!/bin/bash
# run on the GPU box: tools/ab_run.sh '<command>' name1 name2 ...  (variants prebuilt as tools/bin/lib_<name>.so)
CMD="$1"; shift
for rep in 1 2; do for n in "$@"; do cp tools/bin/lib_$n.so dc_rl_amd/csrc/libsustaindc_hip.so; echo "== $n (pass $rep)"; eval "$CMD"; done; done
