set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_diag2; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err ) 2> $O/time20.txt
( time timeout 900 python bench.py --no-pmc --no-cpu-baseline > $O/bench2000.json 2>> $O/bench20.err ) 2> $O/time2000.txt
cat $O/bench20.json; cat $O/time20.txt; cat $O/bench2000.json | cut -c1-600; tail -5 $O/bench20.err
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
