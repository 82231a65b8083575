#!/usr/bin/env python3
"""Summarise a rocprofv3 `--pmc SQ_*` pass: per-launch and per-wavefront averages of the last launches of a kernel.

Usage: python tools/sq_summary.py <pmc_dir> <out.json> [--kernel sdc_dynamics_kernel] [--last 200]
"""
import csv, glob, json, os, sys
from collections import defaultdict

args = [a for a in sys.argv[1:]]
kernel, last = "sdc_dynamics_kernel", 200
if "--kernel" in args:
    i = args.index("--kernel"); kernel = args[i + 1]; del args[i:i + 2]
if "--last" in args:
    i = args.index("--last"); last = int(args[i + 1]); del args[i:i + 2]
d, out = args[0], args[1]
vals = defaultdict(list)
for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(fn, newline="") as f:
        for row in csv.DictReader(f):
            if kernel in row.get("Kernel_Name", ""):
                vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: sum(v[-last:]) / len(v[-last:]) for k, v in sorted(vals.items())}
waves = res.get("SQ_WAVES", 0.0)
if waves:
    res["per_wave"] = {k: res[k] / waves for k in res if k.startswith("SQ_INSTS")}
res["note"] = f"averages over the last {last} launches of {kernel} (rocprofv3 --pmc, one pass)"
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
