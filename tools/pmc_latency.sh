# Latency / cache counters of the step kernel (separate --pmc passes, kernel-trace only):  bash tools/pmc_latency.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pl$i -- python $R/bench.py --pmc-inner --steps 48 --warmup 16 > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/pl1", "/tmp/pl2", "/tmp/pl3"):
    vals = collections.defaultdict(list)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if "sdc_dynamics" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(vals.items()):
        t = v[-32:]
        print(k, round(sum(t) / len(t), 1), "per launch;", round(sum(t) / len(t) / 2176, 2), "per wave")
PY
