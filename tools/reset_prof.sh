# per-kernel time of an episode boundary: rocprofv3 kernel stats over 30 resets of 4096 envs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/rp.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, bench
eng, _, _ = bench.build_engine(4096, 672, 0, seed=1)
for _ in range(30): eng.reset()
torch.cuda.synchronize()
PY
rm -rf /tmp/rp_out; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_out -- python /tmp/rp.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for fn in glob.glob("/tmp/rp_out/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        if "sdc_" in row["Name"]:
            print(row["Name"][:40], "calls", row["Calls"], "avg us", round(float(row["AverageNs"]) / 1e3, 1))
PY
