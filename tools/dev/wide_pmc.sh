# rocprofv3 counter passes over steady-state step launches at N envs (bench.py --pmc-inner: 10 000-step history fill, then 64 launches):
# HBM bytes, SQ issue / wait, memory instructions, L1 / L2 requests, instruction cache, LDS.   bash tools/dev/wide_pmc.sh [N]   (SDC_DEBUG_FLAGS
# is bench.py's: 4096 = lane-per-env kernel off)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-32768}
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_BUSY_CYCLES" \
  "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/wp$i -- python $R/bench.py --pmc-inner --steps 48 --warmup 16 --envs-per-gpu $N > /dev/null 2>&1 || echo "pass $i ($set): rc $?"
done
python - <<PY
import csv, glob, collections
print("N = $N")
for d in sorted(glob.glob("/tmp/wp*")):
    vals = collections.defaultdict(list); name = set()
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if "sdc_dynamics" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"])); name.add(row["Kernel_Name"][:32])
    for k, v in sorted(vals.items()):
        t = v[-32:]
        print(sorted(name), k, round(sum(t) / len(t), 1), "per launch")
PY
rm -rf /tmp/wp*
