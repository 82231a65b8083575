# rocprofv3 passes over tools/dev/wide_prof.py: kernel time, HBM bytes, SQ / cache counters of the step kernel at N envs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-32768}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wt -- python $R/tools/dev/wide_prof.py $N 300 2>&1 | grep "N="
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/wp$i -- python $R/tools/dev/wide_prof.py $N 60 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
for fn in glob.glob("/tmp/wt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "sdc_" in row["Name"]: print(row["Name"][:40], "calls", row["Calls"], "avg ns", row["AverageNs"], "min", row.get("MinNs"), "max", row.get("MaxNs"))
for d in sorted(glob.glob("/tmp/wp*")):
    vals = collections.defaultdict(list)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if "sdc_dynamics" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(vals.items()):
        t = v[-32:]
        print(k, round(sum(t) / len(t), 1), "per launch")
PY
