# SQ counters of the closed-loop kernel (sdc_rollout_actor_kernel, 48 steps per launch), per wavefront and env-step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pa1 -- python $R/tools/actor_run.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d /tmp/pa2 -- python $R/tools/actor_run.py > /tmp/pa2.log 2>&1 || tail -3 /tmp/pa2.log
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/pa1", "/tmp/pa2"):
    vals = collections.defaultdict(list)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if "sdc_rollout_actor" in row["Kernel_Name"]:
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(vals.items()):
        t = v[-4:]
        print("%-28s %10.1f per wavefront and env-step" % (k, sum(t) / len(t) / 2048 / 48))
PY
