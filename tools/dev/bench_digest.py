"""Digest of a bench.py JSON line: python tools/dev/bench_digest.py FILE"""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "steps", d["steps"])
r = d.get("roofline", {})
print("roofline:", {k: r.get(k) for k in ("roofline", "bound", "kernel", "achieved", "frac", "traffic", "hbm_frac", "traffic_over_alg_bytes", "valu_busy_frac", "issue_frac", "kernel_avg_us", "pmc_error")})
s = d.get("secondary", {})
for k, v in s.items():
    if isinstance(v, dict):
        print(k, {x: v.get(x) for x in ("envs", "kernel", "us_per_step", "value", "error") if x in v})
        if "roofline" in v:
            rr = v["roofline"]
            print("   roofline:", {x: rr.get(x) for x in ("bound", "kernel", "frac", "traffic", "hbm_frac", "traffic_over_alg_bytes", "issue_frac", "error", "pmc_error")})
for r in s.get("batch_scan", []):
    print("scan", {x: r.get(x) for x in ("envs", "kernel", "us_per_step", "value", "error") if x in r}, {x: (r[x].get("us_per_step"), r[x].get("value")) for x in ("rollout", "closed_loop") if x in r})
    if "roofline" in r:
        rr = r["roofline"]
        print("   roofline:", {x: rr.get(x) for x in ("bound", "kernel", "frac", "traffic", "hbm_frac", "traffic_over_alg_bytes", "issue_frac", "error", "pmc_error")})
print("cpu_baseline", d.get("cpu_baseline"))
print("rollout", d.get("rollout"), "closed_loop", {k: v for k, v in (d.get("closed_loop") or {}).items() if k != "policy"})
