set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r4a/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4a/tests.log
tail -40 gpurun_out/r4a/tests.log
for n in 4096 16384; do SDC_N=$n SDC_QB_ACTOR=1 tools/ab_run.sh "python tools/qb.py" r3 base licm sink > gpurun_out/r4a/ab_$n.log 2>&1; done
grep -h "==\|step us\|closed" gpurun_out/r4a/ab_*.log
