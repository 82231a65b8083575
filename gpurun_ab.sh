L=dc-rl_amd/csrc/libsustaindc_hip.so
for r in 1 2 3; do
  for v in base new; do cp gpurun_ab_$v.so $L; echo -n "$v: "; timeout 120 python tools/quick_bench.py; done
done
cp gpurun_ab_base.so $L
