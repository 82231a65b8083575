python -m pytest tests/test_gpu_timed_config.py tests/test_gpu_production_sizes.py::test_config3_mixed_racks_4096_production tests/test_gpu_oracle.py -x -q 2>&1 | tail -4
SDC_QB_ACTOR=0 tools/ab_run.sh "python tools/qb.py 2>&1 | tail -1" prev mcfg
cp tools/bin/lib_mcfg.so dc_rl_amd/csrc/libsustaindc_hip.so
python bench.py --mixed-racks --no-pmc --no-rollout --no-secondary --no-cpu-baseline --steps 2000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mixed racks', d['value'], d['ms_per_step'])"
