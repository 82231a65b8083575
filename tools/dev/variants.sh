cd $GRAFT_REPO_ROOT
for f in variants/lib_*.so; do
  cp $f dc_rl_amd/csrc/libsustaindc_hip.so
  echo "== $f"
  for i in 1 2; do python bench.py --no-cpu-baseline --no-pmc --no-rollout | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['kernel_avg_us'], r['kernel_first_entry_to_last_exit_us'])"; done
done
