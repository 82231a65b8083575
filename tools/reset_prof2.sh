cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/rp.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, numpy as np
from dc_rl_amd import dc_config, traces
from dc_rl_amd.engine import SdcEngine
std = float(sys.argv[1]); steps = int(sys.argv[2])
tb = traces.synthetic_tables("ny", 0)
eng = SdcEngine(4096, episode_steps=steps, auto_reset=True, seed=1, weather_noise_std=std)
eng.set_tables(0, tb["W"], tb["C"], tb["T"], tb["WB"]); eng.set_dc_params(0, dc_config.size_datacenter("dc_config.json", 1, 30.0)); eng.assign(0, 0, 170, 190)
for _ in range(30): eng.reset()
torch.cuda.synchronize()
PY
for a in "0.75 672" "0.0 672" "0.75 96"; do rm -rf /tmp/rp_out; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_out -- python /tmp/rp.py $a > /dev/null 2>&1
python - "$a" <<'PY'
import csv, glob, sys
for fn in glob.glob("/tmp/rp_out/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "sdc_" in row["Name"]:
            print(sys.argv[1], "|", row["Name"][:30], "avg us", round(float(row["AverageNs"]) / 1e3, 1))
PY
done
