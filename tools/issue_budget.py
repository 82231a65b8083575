"""profiles/r6_issue_budget.txt: where the 4096-env launch of sdc_dynamics_fast_kernel (the headline) spends its time, from this round's
measurements (gpurun_out/r6 -> profiles/r6_*): rocprofv3 kernel duration, in-kernel wall-clock stamps (tools/wave_timeline.py on the
-DSDC_FAST_DEBUG build), SQ counters per wavefront (tools/pmc_stalls.sh), tools/launch_floor.hip.  usage: python tools/issue_budget.py"""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(R, "profiles", n)
GHZ = 2.4
pmc = {}
for l in open(P("r6_pmc_stalls.txt")):
    m = re.match(r"(\S+) ([\d.]+) per wave", l)
    if m:
        pmc[m.group(1)] = float(m.group(2))
tl = open(P("r6_wave_timeline.txt")).read()
tl1 = open(P("r6_wave_timeline_2048.txt")).read()
g = lambda pat, s: [float(x) for x in re.search(pat, s).groups()]
e_p50, e_p90, e_max = g(r"entry  p50 ([\d.]+) p90 ([\d.]+) max ([\d.]+)", tl)
d_mean, = g(r"dur    mean ([\d.]+)", tl)
d1_mean, = g(r"dur    mean ([\d.]+)", tl1)
le_entry, le_staged, le_dur, le_end = g(r"last-ending wave: entry ([\d.]+) staged ([\d.]+) dur ([\d.]+) end ([\d.]+)", tl)
plain_share, plain_dur = g(r"path 0: share ([\d.]+) dur ([\d.]+)", tl)
st = json.load(open(P("r6_kernel_trace_steady_state.json")))["sdc_dynamics_fast_kernel"]
bench = json.load(open(P("r6_bench_mid_round.json")))
roof = bench["roofline"]
q2us = lambda q: q * 4 / (GHZ * 1e3)
W = 2176.0
env_w = 2048.0
out = []
w = out.append
w("ISSUE BUDGET of the headline launch: sdc_dynamics_fast_kernel, 4096 envs = 2048 env wavefronts (two envs each) + 128 sweep wavefronts,")
w("two env wavefronts per SIMD.  Round 6 measurements, one MI355X box (profiles/r6_pmc_stalls.txt, r6_wave_timeline*.txt,")
w("r6_kernel_trace_steady_state.json, r6_launch_floor.txt, r6_bench_mid_round.json).  us unless noted; SQ counters count 4-cycle quads, 2.4 GHz.")
w("")
w("1. What the driver's clock sees")
w(f"   bench.py per step (HIP events, episode boundaries inside)        {bench['ms_per_step'] * 1e3:6.2f}")
w(f"   of which the amortised episode boundary (reset + features / 672) {bench['episode_boundary']['amortised_us_per_step']:6.2f}")
w(f"   rocprofv3 kernel duration, avg / median of 1500                  {st['avg_us']:6.2f} / {st['median_us']:.2f}   (back-to-back launches hide the launch gap: per step = the kernel)")
w("")
w("2. The kernel's duration, term by term (the launch ends with its LAST wavefront)")
a = st["avg_us"] - le_end
w(f"   a  dispatch start -> first wavefront's first stamp, last wavefront's end -> completion      {a:6.2f}   (= kernel duration - b - c - d)")
w(f"   b  ramp: the last-ending wavefront enters after the first (XCD stagger; p50 {e_p50:.2f}, max {e_max:.2f})       {le_entry:6.2f}")
w(f"   c  staging: its state / inputs arrive (one memory round trip at the start of a launch)        {le_staged - le_entry:6.2f}")
w(f"   d  its life after staging                                                                     {le_dur:6.2f}")
w(f"      = the plain path (no window update / request / arrival: {plain_share * 100:.0f} % of the wavefronts)            {plain_dur:6.2f}")
w(f"      + the tail: what the last-ending wavefront did on top (a key inside a window, a request filed,")
w(f"        a re-centred window taken over; tools/slow_waves.py)                                        {le_dur - plain_dur:6.2f}")
w(f"   a + b + c + d                                                                                 {a + le_end:6.2f}   = the measured {st['avg_us']:.2f} by construction; b + c + d = {le_end:.2f} measured in-kernel")
w("")
w("3. The plain path's 6.1 us is instruction issue, not waiting (per env wavefront; counters / 2176 wavefronts incl. the 128 mostly idle sweep ones)")
any_q, valu_q, sca_q, lds_q = pmc["SQ_ACTIVE_INST_ANY"], pmc["SQ_ACTIVE_INST_VALU"], pmc["SQ_ACTIVE_INST_SCA"], pmc["SQ_ACTIVE_INST_LDS"]
ipw = roof["instructions_per_wavefront"]
w(f"   wave-instructions on the path (SQ_INSTS_*): VALU {ipw['valu']:.0f}, scalar {ipw['salu']:.0f}, LDS {ipw['lds']:.0f}, memory {pmc['SQ_INSTS_VMEM_RD'] + pmc['SQ_INSTS_VMEM_WR']:.0f}, branch {pmc['SQ_INSTS_BRANCH']:.0f}, SMEM {pmc['SQ_INSTS_SMEM']:.0f}")
w(f"   x measured cycles per instruction (SQ_ACTIVE_INST_* / SQ_INSTS_*): VALU {valu_q * 4 / ipw['valu']:.2f}, scalar {sca_q * 4 / ipw['salu']:.2f}, LDS {lds_q * 4 / ipw['lds']:.2f}")
w(f"   = cycles with one of the wavefront's instructions executing: VALU {valu_q * 4:.0f} + scalar {sca_q * 4:.0f} + LDS {lds_q * 4:.0f} + memory / misc {(any_q - valu_q - sca_q - lds_q) * 4:.0f} = {any_q * 4:.0f} cycles = {q2us(any_q):.2f} us")
w(f"   issue stalls (SQ_WAIT_INST_ANY: dependent-issue latency, the port taken by the SIMD's other wavefront) {q2us(pmc['SQ_WAIT_INST_ANY']):.2f} us; parked at s_waitcnt (SQ_WAIT_ANY) {q2us(pmc['SQ_WAIT_ANY']):.2f} us")
w(f"   ONE wavefront per SIMD (2048 envs): life after staging {d1_mean:.2f} -- {q2us(any_q):.2f} of issue + {d1_mean - q2us(any_q):.2f} of dependent-instruction latency and waits nobody fills")
w(f"   TWO per SIMD (4096 envs): life {d_mean:.2f} each; the pair needs the SIMD's issue port for up to 2 x {q2us(any_q):.2f} = {2 * q2us(any_q):.2f} us (no two classes overlapping)")
w(f"   and at least 2 x {q2us(valu_q):.2f} = {2 * q2us(valu_q):.2f} us (VALU alone, everything else hidden under it).  Measured {d_mean:.2f}: the second wavefront's whole {d1_mean:.2f} us")
w(f"   of work costs the first {d_mean - d1_mean:.2f} us -- it already runs in the first one's gaps.  The two wavefronts' active cycles add up to {2 * q2us(any_q):.2f} us inside a")
w(f"   {d_mean:.2f} us window: at least {(2 * q2us(any_q) / d_mean - 1) * 100:.0f} % of them overlap (a scalar / LDS instruction of one under a VALU instruction of the other); every further")
w(f"   idle cycle of the window is matched by one more overlapped one.  bench.py's issue_frac ({roof['issue_frac']:.2f}) is the same sum over the WHOLE launch, ramp and tail included.")
w("")
w("4. What is left, and why the 4096-env kernel is frozen here")
w(f"   * a: {a:.2f} us -- the command processor's; an EMPTY kernel of this grid launched back to back takes 2.9-4.2 us per launch")
w("        (r6_launch_floor.txt), i.e. this launch's fixed part is already overlapped with the previous launch's end.")
w(f"   * b: {le_entry:.2f} us -- the XCDs start a launch up to 1.6 us apart (systematic by XCD pair, r6_wave_timeline.txt rows 'waves 1024-1535');")
w("        the same stagger, larger, on the lane-per-env kernel (profiles/r6_wide_experiments.txt item 2): a property of back-to-back launches")
w("        that write memory, not of this kernel's code.")
w(f"   * c: {le_staged - le_entry:.2f} us -- one round trip to HBM / the Infinity Cache with every wavefront asking at once; one round trip is the minimum.")
w(f"   * d plain: {plain_dur:.2f} us = {q2us(any_q):.2f} us of this wavefront's issue + the neighbour's; shortening it means REMOVING instructions:")
w(f"        1 % of the path ({(ipw['valu'] + ipw['salu'] + ipw['lds']) / 100:.0f} instructions) is worth ~0.06 us.  Rounds 2-5 took the path from ~2500 to {ipw['valu'] + ipw['salu'] + ipw['lds']:.0f} instructions;")
w("        round 5's one remaining re-ordering experiment lost (DESIGN.md 4.7).")
w(f"   * d tail: {le_dur - plain_dur:.2f} us -- VERDICT r5 item 2 proposed moving the whole-window work of a step (a key inside a window: ~5 % of")
w("        env-steps = one in five two-env wavefronts every step) to the next launch's sweep workgroups.  Those updates change what the")
w("        NEXT step reads (the window's keys, its first / last key, the ranks): deferring them needs the one-step replay machinery of the")
w("        deferred re-centrings for EVERY window update -- ~400 a step instead of ~26 -- on 128 sweep wavefronts that already run 5 us each:")
w("        the sweeps would become the launch's tail.  Not attempted; the tail without the 100 slowest wavefronts is still")
span100, = g(r"span without the 100 longest waves: ([\d.]+)", tl)
w(f"        {span100:.2f} us of first-entry-to-last-exit against {le_end:.2f}: at most {le_end - span100:.2f} us (6 %) is there to win.")
w("   The rate keeps rising with the batch instead: 1.32-1.40 G env-steps/s at 32 768 envs on the lane-per-env kernel (3.7-3.9 x the headline).")
open(P("r6_issue_budget.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
