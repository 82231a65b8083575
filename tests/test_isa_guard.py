"""The ISA properties the step's performance rests on, checked on the CPU (hipcc cross-compiles gfx950 without a GPU).

A compiler bump or an innocent edit regresses these silently -- the kernels stay correct and get slower:
  * no private segment (scratch) in ANY step / rollout kernel: a kernel with scratch starts its wavefronts 0.6 us later
    (DESIGN.md section 4 "Scratch slows the dispatcher"), and a spill inside a multi-step kernel's loop is paid every step;
  * <= 168 VGPRs for the single-step kernels and the four-envs-per-wavefront rollout kernel (three resident wavefronts per
    SIMD: two env wavefronts + room for a sweep wavefront / three quad wavefronts);
  * the env's three actions are requested by ONE hand-issued `global_load_dwordx3` inside the first burst of loads (before
    any wait for memory: no second round trip), and nothing reads its destination before an `s_waitcnt vmcnt(0)`
    (the inline asm is invisible to the compiler's wait-count insertion: sdc_step.hip "requested FIRST and by hand").
  * the episode boundary's two kernels keep their residency: `sdc_reset_kernel` <= 128 VGPRs (four wavefronts per SIMD = all
    4096 resets of the timed configuration in flight at once: the kernel is VALU-issue bound and ends with its last wavefront)
    and `sdc_features_kernel` <= 168 (three per SIMD; its LDS windows are shared by the four wavefronts of an env), no scratch.
  * the lane-per-env kernel (sdc_wide.hip) has no scratch, at most 40 KB of LDS per workgroup and two wavefronts per SIMD.
The step kernels' translation units (sdc_step.hip, sdc_rollout.hip, sdc_wide.hip) are compiled once, with the production flags of
dc_rl_amd/_lib.py."""
import os
import re
import subprocess
import tempfile

import pytest

from dc_rl_amd import _lib as L

STEP_KERNELS = ["sdc_dynamics_kernel", "sdc_dynamics_fast_kernel", "sdc_dynamics_quad_kernel"]
LOOP_KERNELS = ["sdc_rollout_kernel", "sdc_rollout_fast_kernel", "sdc_rollout_quad_kernel", "sdc_rollout_actor_kernel",
                "sdc_rollout_actor_quad_kernel"]
VGPR_CAP_3_WAVES = 168


@pytest.fixture(scope="module")
def compiled():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    flags = [f for f in L.HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory() as td:
        asm, err = "", ""
        for src in ("sdc_step.hip", "sdc_rollout.hip", "sdc_wide.hip"):      # (the step kernels' three translation units)
            out = os.path.join(td, src[:-4] + ".s")
            r = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", src, "-o", out],
                               cwd=L.CSRC, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            asm += open(out).read()
            err += r.stderr
    usage = {}
    name = None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    return asm, usage


def _usage_of(src):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    flags = [f for f in L.HIPCC_FLAGS if f != "-shared"]
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([hipcc] + flags + ["-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", src, "-o", os.path.join(td, "o.o")],
                           cwd=L.CSRC, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    u = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m:
            u[m.group(1).strip()] = int(m.group(2))
    return u


@pytest.mark.parametrize("src,vgprs,waves", [("sdc_reset.hip", 128, 4), ("sdc_features.hip", 168, 3)])
def test_episode_boundary_kernels_keep_their_residency(src, vgprs, waves):
    u = _usage_of(src)
    assert u["ScratchSize"] == 0 and u["VGPRs Spill"] == 0, (src, u)
    assert u["VGPRs"] <= vgprs and u["Occupancy"] >= waves, (src, u)
    print(src, u["VGPRs"], u["Occupancy"], u.get("LDS Size"))


def test_no_scratch_and_register_budget(compiled):
    _, usage = compiled
    for k in STEP_KERNELS + LOOP_KERNELS:
        assert k in usage, (k, sorted(usage))
        u = usage[k]
        assert u["ScratchSize"] == 0 and u["VGPRs Spill"] == 0, (k, u)
    for k in STEP_KERNELS + ["sdc_rollout_quad_kernel"]:
        assert usage[k]["VGPRs"] <= VGPR_CAP_3_WAVES and usage[k]["Occupancy"] >= 3, (k, usage[k])
    w = usage["sdc_dynamics_wide_kernel"]
    # one lane per env: two wavefronts per 64 envs, 40 KB of LDS per workgroup (four per CU), two wavefronts per SIMD (<= 256 registers)
    assert w["ScratchSize"] == 0 and w["VGPRs Spill"] == 0 and w["LDS Size"] <= 40960 and w["VGPRs"] + w.get("AGPRs", 0) <= 256 and w["Occupancy"] >= 2, w
    print({k: (usage[k]["VGPRs"], usage[k]["Occupancy"]) for k in STEP_KERNELS + LOOP_KERNELS + ["sdc_dynamics_wide_kernel"]})


def _kernel_body(asm, name):
    lines = asm.split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return [l.strip() for l in lines[i0 + 1:i1] if l.strip() and not l.strip().startswith((";", ".loc", ".Ltmp", ".cfi"))]


def _vregs(operand_text):
    """VGPR numbers named in an operand string: v12, v[4:7]."""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        regs |= set(range(int(a), int(b) + 1))
    regs |= {int(x) for x in re.findall(r"\bv(\d+)\b", operand_text)}
    return regs


@pytest.mark.parametrize("kernel", STEP_KERNELS)
def test_actions_are_requested_first_and_waited_for(compiled, kernel):
    asm, _ = compiled
    body = _kernel_body(asm, kernel)
    loads = [i for i, l in enumerate(body) if l.startswith("global_load_dwordx3")]
    assert loads, "the hand-issued action load is gone"
    first_wait = next(i for i, l in enumerate(body) if l.startswith("s_waitcnt") and "vmcnt" in l)
    i = loads[0]
    assert i < first_wait, (kernel, "the action load is issued behind a wait for memory: a second round trip", i, first_wait)
    # (the state record is requested in the same burst, behind the actions: memory returns loads in order)
    assert any(l.startswith(("global_load_dwordx2", "global_load_dwordx4")) for l in body[i + 1:first_wait]), kernel
    dest = _vregs(body[i].split(",")[0])
    assert len(dest) == 3, body[i]
    labels = {l[:-1]: j for j, l in enumerate(body) if l.endswith(":")}
    waited, j, steps = False, i + 1, 0
    while j < len(body) and steps < 100000:       # the likely path: fall through conditional branches, follow unconditional ones
        l = body[j]
        j, steps = j + 1, steps + 1
        if l.startswith("s_branch"):
            j = labels[l.split()[1]]
            continue
        if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            waited = True
        op = l.split(None, 1)
        # a READ of a destination register: a source operand (everything behind the first operand; a store reads them all)
        src = op[1] if len(op) == 2 and "_store" in op[0] else (op[1].split(",", 1)[1] if len(op) == 2 and "," in op[1] else "")
        if not l.startswith(("s_waitcnt", ".LBB")) and _vregs(src) & dest:
            assert waited, (kernel, "the actions are read before a vmcnt(0) wait", l)
            break
    else:
        pytest.fail("the actions are never read")
