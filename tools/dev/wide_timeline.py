"""Timeline of the lane-per-env kernel's two wavefronts (a -DSDC_WIDE_STAMPS build: tools/dev/ab_wide.sh): wall-clock stamps of lane 0."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench
N = int(sys.argv[1])
# steady state as bench.py reaches it: 10 000-step history fill from the synthetic traces (injected rings of i.i.d. values make a step's
# energy land inside the rank windows far more often than a real history does)
e, tb, params = bench.build_engine(N, 672, 0, seed=1234, debug_flags=2048)
g = torch.Generator(device="cpu").manual_seed(1234)
acts = torch.randint(0, 3, (64, N, 3), dtype=torch.int32, generator=g).to("cuda:0")
e.reset()
for t in range(10300): e.step(acts[t % 64])
assert e.last_step_kernel() == "sdc_dynamics_wide_kernel", e.last_step_kernel()
acc = []
for t in range(100):
    e.step(acts[t % 64]); acc.append(e.info[::64, :16].cpu().numpy().astype(np.int64).copy())
a = np.stack(acc)                                  # [steps, workgroups, 16] ticks mod 2^24
t0 = a[..., 0:1]
d = ((a - t0) % (1 << 24)) / 100.0                 # us after the dynamics wavefront's entry
names = ["D entry", "D first blocks in LDS (barrier 1)", "D load shifting done", "D rack model done", "D energy known (hand-over)", "D state out starts",
         "D outputs start", "D info staged (barrier 3)", "R headers in LDS (barrier 1)", "R arrivals done, at barrier 2", "R past barrier 2: updates start",
         "R window updates done", "R moments done", "R committed", "R header out + fallback done", "R info out"]
print(f"N={N}: mean / p90 us after the dynamics wavefront's entry (100 launches x {a.shape[1]} workgroups)")
for i, nm in enumerate(names): print(f"  {nm:48s} {d[..., i].mean():7.2f} {np.percentile(d[..., i], 90):7.2f}")
st = a[..., 0]
sp = ((st - st.min(1, keepdims=True)) % (1 << 24)) / 100.0
print("  workgroup entry spread within a launch (us): p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(sp, [50, 90, 100])))
en = ((a[..., 15] - st.min(1, keepdims=True)) % (1 << 24)) / 100.0
print("  last stamp after the launch's first entry (us): p50 %.2f p90 %.2f max(mean over launches) %.2f" % (np.percentile(en, 50), np.percentile(en, 90), en.max(1).mean()))
# the workgroup that ends LAST in each launch (it sets the kernel's duration): its own timeline, and where it entered
last = en.argmax(1)
L = np.arange(a.shape[0])
dl = d[L, last]                                    # [launch, 16]
print("  the last-ending workgroup of each launch: entry after the launch's first entry %.2f us; its stamps (mean / p90):" % sp[L, last].mean())
for i, nm in enumerate(names): print(f"    {nm:46s} {dl[:, i].mean():7.2f} {np.percentile(dl[:, i], 90):7.2f}")
seg = np.diff(d[..., [0, 1, 2, 3, 4]], axis=-1), np.diff(d[..., [10, 11, 12, 13, 14, 15]], axis=-1)
print("  segment durations, all workgroups vs the last-ending one (us): D loads / load shifting / rack / HVAC+battery | R updates / moments / commit / header out + fallback / info")
print("    all : " + " ".join("%.2f" % x for x in np.concatenate([seg[0].mean((0, 1)), seg[1].mean((0, 1))])))
print("    last: " + " ".join("%.2f" % x for x in np.concatenate([seg[0][L, last].mean(0), seg[1][L, last].mean(0)])))
print("    p99 : " + " ".join("%.2f" % x for x in np.concatenate([np.percentile(seg[0], 99, axis=(0, 1)), np.percentile(seg[1], 99, axis=(0, 1))])))
rack = seg[0][..., 2]; upd = seg[1][..., 0]; tot = d[..., 15]
print("  rack segment by launch (mean over workgroups), first 12 launches: " + " ".join("%.2f" % x for x in rack.mean(1)[:12]) + " ; spread inside a launch (p99 - p50): %.2f" % (np.percentile(rack, 99, axis=1) - np.percentile(rack, 50, axis=1)).mean())
mx = en.max(1)
print("  the launch's last stamp after its first entry (us): median over launches %.2f, p90 %.2f" % (np.median(mx), np.percentile(mx, 90)))
wgm = rack.mean(0)
o = np.argsort(-wgm)[:8]
print("  slowest rack by workgroup index (mean over launches):", [(int(i), round(float(wgm[i]), 2)) for i in o], "overall std across workgroups of the mean %.2f" % wgm.std())
ls_seg = seg[0][..., 1]; hv = seg[0][..., 3]
slow = rack > np.percentile(rack, 95)
print("  workgroup-launches in the top 5 %% of rack time: rack %.2f, load shifting %.2f (all %.2f), HVAC+battery %.2f (all %.2f), R updates %.2f (all %.2f)" % (rack[slow].mean(), ls_seg[slow].mean(), ls_seg.mean(), hv[slow].mean(), hv.mean(), upd[slow].mean(), upd.mean()))
# entry of the workgroups by position in the grid (env workgroup index = blockIdx.x - sweep blocks; first_pair_of_block maps it to envs):
# a[..., 0] is indexed by ENV block; print by env block in groups of 32
grp = 32
print("  entry after the launch's first entry (us), mean by env block / %d:" % grp, " ".join("%.1f" % sp[:, i:i + grp].mean() for i in range(0, sp.shape[1], grp)))
print("  entry percentiles over workgroups (us): " + " ".join("p%d %.2f" % (q, np.percentile(sp, q)) for q in (10, 25, 50, 75, 90, 99)))
