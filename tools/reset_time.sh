# per-kernel time of the episode boundary (rocprofv3 kernel trace over 12 resets of 4096 envs, 672-step episodes)
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
rm -rf /tmp/rt_out; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt_out -- python /tmp/rp.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for fn in glob.glob("/tmp/rt_out/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "sdc_" in row["Name"]:
            print("  ", row["Name"][:30], "avg us", round(float(row["AverageNs"]) / 1e3, 1), "min", round(float(row["MinNs"]) / 1e3, 1))
PY
