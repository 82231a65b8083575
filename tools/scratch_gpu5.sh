for n in 8192 12288 16384; do SDC_N=$n SDC_QB_ACTOR=1 tools/ab_run.sh "python tools/qb.py 2>&1 | tail -3" ll1 ll2; done
