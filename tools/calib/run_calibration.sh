#!/bin/bash
# rocprofv3 counter passes over tools/bin/counter_calibration (known byte counts in the lane-per-env kernel's access patterns) ->
# gpurun_out/r6/counter_calibration.txt.  Run on the GPU box: bash tools/calib/run_calibration.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6
mkdir -p $OUT
$R/tools/bin/counter_calibration 1 > /tmp/cal_truth.txt || exit 1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_WR_UNCACHED_32B_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/cal$i -- $R/tools/bin/counter_calibration 2 > /tmp/cal_pass$i.log 2>&1 || echo "pass $i ($set): rc $?"
done
python3 $R/tools/calib/summarize_calibration.py /tmp/cal_truth.txt /tmp/cal[0-9]* > $OUT/counter_calibration.txt
cat $OUT/counter_calibration.txt
rm -rf /tmp/cal[0-9]*
