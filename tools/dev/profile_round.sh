set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r1_v9
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 2000 > $O/bench_under_rocprof.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/sq -- python $R/bench.py --steps 300 --no-cpu-baseline > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python $R/bench.py --steps 300 --no-cpu-baseline > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python $R/bench.py --steps 300 --no-cpu-baseline > $O/write.log 2>&1
cd $R
python tools/sq_summary.py $O/sq $O/sq_counters.json > /dev/null
python tools/pmc_summary.py $O/fetch $O/write; cp profiles/hbm_traffic.json $O/hbm_traffic.json
python tools/trace_summary.py $O/trace --last 1500 --out $O/kernel_trace_steady_state.json 2>&1 | tail -3
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
# drop the bulky raw traces, keep summaries
find $O -name "*kernel_trace.csv" -size +1M -delete; find $O -name "*counter_collection.csv" -size +1M -delete
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --mixed-racks --no-cpu-baseline > $O/bench_mixed_racks.json 2>> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
ls $O
