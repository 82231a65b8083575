#!/usr/bin/env python3
"""Walk the likely path of a kernel in hipcc's -S output: fall through conditional branches (the compiler lays the
expected path out as the fall-through), follow unconditional ones, stop at s_endpgm.  Prints an instruction-class
histogram and, with --dump, the walked instructions with their source-line markers.
usage: tools/isa_hotpath.py step.s sdc_dynamics_kernel [--dump out.txt] [--take LABEL ...]   (--take: conditional branches
to LABEL are taken instead)"""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    take = set()
    if "--take" in sys.argv:
        take = set(sys.argv[sys.argv.index("--take") + 1:])
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    labels = {}
    for i in range(start, len(lines)):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
        if lines[i].startswith(".Lfunc_end"):
            break
    FILES = {}
    for l in lines:
        m = re.match(r'^\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            FILES[m.group(1)] = m.group(2).split("/")[-1]
    cur_loc = ("?", "0")
    exiting = None
    branches = []
    i = start + 1
    hist = collections.Counter()
    cls = collections.Counter()
    out = []
    seen_jump = 0
    n = 0
    visited = collections.Counter()
    while True:
        l = lines[i]
        s = l.strip()
        i += 1
        if s.startswith(".loc"):
            f = s.split()
            cur_loc = (FILES.get(f[1], f[1]), f[2])
            continue
        if not s or s.startswith(";") or s.startswith(".") and not s.startswith(".LBB"):
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            if dump:
                out.append(l)
            continue
        op = s.split()[0]
        n += 1
        hist[op] += 1
        if op.startswith("v_"):
            c = "valu"
        elif op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_setpc") or op.startswith("s_swappc"):
            c = "branch"
        elif op.startswith("s_waitcnt"):
            c = "waitcnt"
        elif op.startswith("s_nop"):
            c = "nop"
        elif op.startswith("s_"):
            c = "salu"
        elif op.startswith("ds_"):
            c = "lds"
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            c = "vmem"
        else:
            c = "other"
        cls[c] += 1
        if dump:
            out.append("%-90s ; %s:%s" % (l.rstrip(), cur_loc[0], cur_loc[1]))
        if op == "s_endpgm":
            break
        if op == "s_branch":
            tgt = s.split()[1]
            visited[tgt] += 1
            if visited[tgt] > 8:
                out.append("; ---- loop detected, stop")
                break
            if labels[tgt] < i:          # back edge: one trip through the loop body, then leave by its first exit
                exiting = (labels[tgt], i)
            i = labels[tgt]
        elif op.startswith("s_cbranch"):
            tgt = s.split()[1]
            t = tgt in take
            if exiting and not (exiting[0] <= labels[tgt] <= exiting[1]) and exiting[0] <= i <= exiting[1]:
                t = True
                exiting = None
            if labels[tgt] < i and not t and tgt not in take:    # conditional back edge (loop latch): not taken = one trip
                pass
            branches.append((n, op, tgt, cur_loc, t))
            if t:
                i = labels[tgt]
        if n > 20000:
            break
    print("instructions walked:", n)
    print(dict(cls))
    for k, v in hist.most_common(45):
        print("%5d %s" % (v, k))
    if dump:
        open(dump, "w").write("\n".join(out))
        open(dump + ".branches", "w").write("\n".join("%5d %-18s %-12s %s:%s %s" % (a, b, c, d[0], d[1], "TAKEN" if e else "")
                                                       for a, b, c, d, e in branches))


main()
