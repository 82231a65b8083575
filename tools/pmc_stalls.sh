# usage: bash tools/pmc_stalls.sh            (4096 envs: the two-env kernel, 2048 + 128 wavefronts per launch)
#        SDC_PMC_ENVS=16384 bash tools/pmc_stalls.sh   (the four-env kernel: 4096 + 128 wavefronts per launch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=${SDC_PMC_ENVS:-4096}
export SDC_PMC_WAVES=$(( E >= 5636 ? E / 4 + 128 : E / 2 + 128 ))
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/p1 -- python $R/bench.py --pmc-inner --steps 48 --warmup 16 --envs-per-gpu $E > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG --kernel-trace --output-format csv -d /tmp/p2 -- python $R/bench.py --pmc-inner --steps 48 --warmup 16 --envs-per-gpu $E > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, os
W = float(os.environ.get('SDC_PMC_WAVES', '2176'))
for d in ("/tmp/p1", "/tmp/p2"):
    vals = collections.defaultdict(list)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if "sdc_dynamics" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(vals.items()):
        t = v[-32:]
        print(k, round(sum(t) / len(t) / W, 1), "per wave")
PY
