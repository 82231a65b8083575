"""Truth table of counter_calibration + the rocprofv3 counter passes -> one row per access pattern: what each counter reports per
launch against the bytes the launch requested / the 32-, 64- and 128-byte blocks it touched."""
import collections, csv, glob, sys

truth = []
for line in open(sys.argv[1]):
    if line.startswith("#") or not line.strip():
        continue
    pat, rest = line.split(" ", 1)
    kern, req, t32, t64, t128 = rest.rsplit(" ", 4)
    truth.append((pat, kern.strip(), float(req), float(t32), float(t64), float(t128)))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))


def find(kern):
    for name, c in vals.items():
        if kern in name:
            return {k: sum(v) / len(v) for k, v in c.items()}
    return {}


MB = 1e6
print("counter calibration on the lane-per-env kernel's access patterns (per launch; sizes in MB = 1e6 bytes)")
print("FETCH_SIZE / WRITE_SIZE are reported in KiB (x 1024 below, NO other correction); RDREQ / WRREQ are request counts")
hdr = ["pattern", "requested", "touched32", "touched64", "touched128", "FETCH_SIZE", "WRITE_SIZE", "RDREQ", "RDREQ_32B", "WRREQ", "WRREQ_64B",
       "fetch/req", "fetch/t64", "fetch/t128", "write/req", "write/t32", "write/t64"]
print(" | ".join(hdr))
for pat, kern, req, t32, t64, t128 in truth:
    c = find(kern)
    f = c.get("FETCH_SIZE", float("nan")) * 1024
    w = c.get("WRITE_SIZE", float("nan")) * 1024
    g = lambda k: c.get(k, float("nan"))
    is_read = pat[0] in "ABCDH"
    cols = [pat, f"{req / MB:.2f}", f"{t32 / MB:.2f}", f"{t64 / MB:.2f}", f"{t128 / MB:.2f}", f"{f / MB:.2f}", f"{w / MB:.2f}",
            f"{g('TCC_EA0_RDREQ_sum'):.0f}", f"{g('TCC_EA0_RDREQ_32B_sum'):.0f}", f"{g('TCC_EA0_WRREQ_sum'):.0f}", f"{g('TCC_EA0_WRREQ_64B_sum'):.0f}"]
    if is_read:
        cols += [f"{f / req:.3f}", f"{f / t64:.3f}", f"{f / t128:.3f}", "-", "-", "-"]
    else:
        cols += ["-", "-", "-", f"{w / req:.3f}", f"{w / t32:.3f}", f"{w / t64:.3f}"]
    print(" | ".join(cols))
print()
print("other counters per launch:")
for pat, kern, *_ in truth:
    c = find(kern)
    print(pat, {k: round(v, 1) for k, v in sorted(c.items()) if k not in ("FETCH_SIZE", "WRITE_SIZE")})
