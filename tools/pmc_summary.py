#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/hbm_traffic.json.

Usage: python tools/pmc_summary.py <fetch_dir> <write_dir> [--kernel sdc_dynamics_kernel] [--last 200]

Each dir holds the `*_counter_collection.csv` of one `rocprofv3 --pmc X --kernel-trace` pass (FETCH_SIZE and
WRITE_SIZE need separate passes: TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2).  Corrections follow
/opt/skills/guides/MI355X_MICROARCH.md section HBM: the counters are in KiB, and on gfx950 FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) coalesced streaming read, so the fetch side is doubled.
WRITE_SIZE is uncalibrated there; it is reported as is.
"""
import csv
import glob
import json
import os
import sys


def per_kernel(dirname, counter, kernel, last):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {dirname}")
    vals = []
    with open(files[0], newline="") as f:
        for row in csv.DictReader(f):
            if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"{counter} for {kernel} not found in {files[0]}")
    tail = vals[-last:]
    return sum(tail) / len(tail), len(vals)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kernel = "sdc_dynamics_kernel"
    last = 200
    for i, a in enumerate(sys.argv):
        if a == "--kernel":
            kernel = sys.argv[i + 1]
        if a == "--last":
            last = int(sys.argv[i + 1])
    args = [a for a in args if a not in (kernel, str(last))]
    fetch_dir, write_dir = args[0], args[1]
    fetch_kib, nf = per_kernel(fetch_dir, "FETCH_SIZE", kernel, last)
    write_kib, nw = per_kernel(write_dir, "WRITE_SIZE", kernel, last)
    out = {
        "kernel": kernel,
        "launches_seen": nf,
        "averaged_over_last": min(last, nf),
        "FETCH_SIZE_KiB_raw": fetch_kib,
        "WRITE_SIZE_KiB_raw": write_kib,
        "fetch_bytes_per_launch": fetch_kib * 1024 * 2,     # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        "write_bytes_per_launch": write_kib * 1024,
        "hbm_bytes_per_launch": fetch_kib * 1024 * 2 + write_kib * 1024,
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (wide coalesced reads report half); WRITE_SIZE uncalibrated; "
                "Infinity-Cache hits are counted by these fabric-side counters",
    }
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    with open(os.path.join(root, "profiles", "hbm_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
