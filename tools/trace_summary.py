#!/usr/bin/env python3
"""Steady-state per-kernel durations from a rocprofv3 --kernel-trace CSV: average over the LAST `--last` launches
of each kernel (bench.py first runs 10 000 history-fill steps with shorter rings, which would bias --stats).
Usage: python tools/trace_summary.py <dir with *_kernel_trace.csv> [--last 1000] [--out profiles/x.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 1000
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = defaultdict(list)
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if name.startswith("sdc_"):
                dur[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    res = {}
    for k, v in dur.items():
        tail = v[-last:] if k != "sdc_reset_kernel" else v
        tail_sorted = sorted(tail)
        res[k] = {"calls_total": len(v), "averaged_over_last": len(tail), "avg_us": sum(tail) / len(tail) / 1e3,
                  "median_us": tail_sorted[len(tail) // 2] / 1e3, "min_us": tail_sorted[0] / 1e3,
                  "max_us": tail_sorted[-1] / 1e3}
    print(json.dumps(res, indent=1))
    if out:
        with open(out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
