cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_multi_device.py tests/test_gpu_distributed.py::test_rccl_world_size_1 tests/test_gpu_surface.py tests/test_gpu_actor.py -x -q -s > gpurun_out/r4b/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4b/tests.log
tail -30 gpurun_out/r4b/tests.log
timeout 600 python tools/harl_loop_rate.py 48 512 4096 > gpurun_out/r4b/harl_loop.txt 2>&1; cat gpurun_out/r4b/harl_loop.txt
timeout 600 python bench.py --gpus 2 --single-process --devices 0,0 --steps 200 --warmup 20 > gpurun_out/r4b/bench_sp2.json 2> gpurun_out/r4b/bench_sp2.err; tail -3 gpurun_out/r4b/bench_sp2.err; cat gpurun_out/r4b/bench_sp2.json
timeout 600 python bench.py --gpus 1 --single-process --steps 200 --warmup 20 > gpurun_out/r4b/bench_sp1.json 2> gpurun_out/r4b/bench_sp1.err; cat gpurun_out/r4b/bench_sp1.json
