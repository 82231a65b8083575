set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_diag
O=gpurun_out/r2_diag
./tools/dev/dispatch_rate > $O/dispatch.txt 2>&1
timeout 300 python tools/launch_spread.py > $O/spread.txt 2>&1
timeout 300 python tools/wave_phases.py > $O/phases.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
timeout 300 python bench.py --steps 2000 --no-cpu-baseline > $O/bench2000.json 2>> $O/bench20.err
cat $O/dispatch.txt $O/spread.txt $O/phases.txt; cut -c1-400 $O/bench20.json; cut -c1-400 $O/bench2000.json
