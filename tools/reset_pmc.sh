# PMC view of the episode boundary's two kernels (sdc_reset_kernel, sdc_features_kernel): is the noise walk VALU-issue bound?
# usage (GPU box): bash tools/reset_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/rp.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
tb = traces.synthetic_tables("ny", 0)
eng = SdcEngine(4096, episode_steps=672, auto_reset=True, seed=1)
eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"]); eng.set_dc_params(0, dc_config.size_datacenter("dc_config.json", 1, 30.0)); eng.assign(0, 0, 170, 190)
for _ in range(12): eng.reset()
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/q1 -- python /tmp/rp.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_WAVES --kernel-trace --output-format csv -d /tmp/q2 -- python /tmp/rp.py > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/q3 -- python /tmp/rp.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/q4 -- python /tmp/rp.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/q1", "/tmp/q2", "/tmp/q3", "/tmp/q4"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            for k in ("sdc_reset", "sdc_features"):
                if k in row["Kernel_Name"]:
                    vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in vals:
        for c, v in sorted(vals[k].items()):
            t = v[-8:]
            if c in ("FETCH_SIZE", "WRITE_SIZE"):   # KiB per launch (FETCH_SIZE x 2 on gfx950: MI355X_MICROARCH.md, HBM section)
                print(k, c, round(sum(t) / len(t) * (2 if c == "FETCH_SIZE" else 1) * 1024 / 1e6, 1), "MB per launch" + (" (counter x 2)" if c == "FETCH_SIZE" else ""))
            else:
                print(k, c, round(sum(t) / len(t) / 4096, 1), "per wavefront (4096 per launch)")
    for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for row in csv.DictReader(open(fn)):
            dur[row["Kernel_Name"][:28]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            if "sdc_" in k: print(d, k, "us:", round(sum(v[-8:]) / len(v[-8:]), 1))
PY
