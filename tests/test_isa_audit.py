"""The shipped kernel's hand-scheduled pieces, checked in the gfx950 assembly (cross-compiled here, no GPU needed).

cbet_trace_window.hip issues the per-step record gather from inline assembly and waits for it with a counted
`s_waitcnt vmcnt(N)`, so the compiler does not know that the destination registers are written asynchronously
between the two.  That is only correct if it leaves the record where the load puts it: both assembly blocks print
the registers they were given (CBET_RECORD_ISSUE / CBET_RECORD_WAIT comments) and this test requires, for every
instantiation of the kernel, that they all name the same registers and that nothing in between touches them.
It also pins the properties DESIGN.md quotes: no fused multiply-add in the kernels (one IEEE operation per reference
statement), native fp64 atomics (no compare-and-swap loops), no scratch, the LDS size that gives 16 waves per CU --
and the gfx940 / gfx950 wait states in front of every inline-assembly instruction that reads a scalar register (the
compiler tracks those hazards for its own instructions only; round 4's audited build faulted on one)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "cbet_raytracing_3d_amd", "csrc")


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    from cbet_raytracing_3d_amd import build
    out = tmp_path_factory.mktemp("isa") / "window.s"
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [build.hipcc()] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                                     "-o", str(out), os.path.join(CSRC, "cbet_trace_window.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_ZN4cbet\S*k_trace_window\S*):[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
        kernels[m.group(1)] = m.group(2)
    return kernels


def _regs(spec):
    """'v[6:9]' -> {6,7,8,9}; 'v12' -> {12}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", spec)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(spec[1:])}


def _touched(line):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", line):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def test_all_instantiations_present(listing):
    names = "\n".join(listing)
    # WZ 16 x {plain, gain hooks, energy-field pass} and the four-component field pass at WZ 8, each in the
    # compiled-in common case and the generic (run-time absorption flag, 64-bit indexing) form
    # ... and each of those with and without the window diagnostics (cbet_params.window_stats)
    for inst in ("ILi16ELb0ELi0E", "ILi16ELb1ELi0E", "ILi16ELb0ELi1E", "ILi16ELb1ELi1E", "ILi16ELb0ELi2E", "ILi16ELb1ELi2E",
                 "ILi8ELb0ELi4E", "ILi8ELb1ELi4E"):
        for stats in ("Lb0EEE", "Lb1EEE"):
            assert inst + stats in names, inst + stats
    assert len(listing) == 16


def test_in_flight_record_is_never_moved(listing):
    for name, body in listing.items():
        lines = body.splitlines()
        marks = [(i, l.split()) for i, l in enumerate(lines) if "CBET_RECORD_" in l]
        issues = [(i, w) for i, w in marks if "CBET_RECORD_ISSUE" in w]
        waits = [(i, w) for i, w in marks if "CBET_RECORD_WAIT" in w]
        assert len(issues) == 2 and len(waits) == 2, name     # before the loop, and in it
        regs = {tuple(w[-2:]) for _, w in marks}
        assert len(regs) == 1, (name, regs)                   # the same registers at every issue and every wait
        record = _regs(marks[0][1][-2]) | _regs(marks[0][1][-1])
        assert len(record) == 8
        # straight-line stretches: issue -> wait before the loop; in the loop the wait follows the issue in text
        # order too (one basic-block chain) -- nothing between them may name the record's registers
        for (i0, _), (i1, _) in zip(issues, waits):
            assert i1 > i0, name
            for l in lines[i0 + 1:i1]:
                code = l.split(";")[0]
                if not code.strip() or code.strip().endswith(":"):
                    continue
                assert not (_touched(code) & record), (name, l)


def test_every_write_back_atomic_is_counted():
    """record_wait(vmcnt(N <= pend)) is only right while every vector-memory instruction issued after a gather is
    counted in WaveCounters::pend: every global_add / grid_add call site of the shipped kernel sits behind a `wc.pend +=` of its
    own (the four-component field pass waits with vmcnt(0) and is exempt).  The file refuses targets other than gfx950
    (one in-order vmcnt shared by loads, stores and atomics)."""
    src = open(os.path.join(CSRC, "cbet_trace_window.hip")).read().splitlines()
    # (grid_add / grid_add_far: global_add through the grid handle; their own bodies -- `g.p` -- are not call sites)
    sites = [i for i, l in enumerate(src) if re.search(r"\b(global_add|grid_add|grid_add_far)\(a,", l) and "__device__" not in l and "g.p" not in l]
    assert len(sites) >= 12
    helper = [i for i, l in enumerate(src) if "auto hbm_add8 = " in l]          # eight adds of a lane outside both boxes:
    assert len(helper) == 1                                                      # counted where the helper is CALLED
    for i in sites:
        window = "\n".join(src[max(0, i - 12):i + 1])
        assert ("wc.pend +=" in window or "comp_stride + own_node" in window or
                helper[0] < i <= helper[0] + 10), (i + 1, src[i])
    calls = [i for i, l in enumerate(src) if "hbm_add8(" in l and "auto hbm_add8" not in l]
    assert len(calls) == 2
    for i in calls:
        assert "wc.pend +=" in "\n".join(src[max(0, i - 4):i + 1]), (i + 1, src[i])
    text = "\n".join(src)
    assert "#error" in text and "__gfx950__" in text


def test_native_fp64_atomics(listing):
    """HBM and LDS adds are the native instructions, never compare-and-swap loops or flat atomics."""
    for name, body in listing.items():
        assert "global_atomic_cmpswap" not in body and "flat_atomic" not in body and "ds_cmpst" not in body, name
        assert "global_atomic_add_f64" in body and "ds_add_f64" in body, name


def test_resources_of_the_headline_instance(listing):
    (name, body), = [(n, b) for n, b in listing.items() if "ILi16ELb0ELi0ELb0E" in n]
    meta = dict(re.findall(r"\.amdhsa_(\w+)\s+(\S+)", body))
    assert int(meta["group_segment_fixed_size"]) == 10240          # 16 waves per CU of 160 KB
    assert int(meta["private_segment_fixed_size"]) == 0            # no scratch
    assert int(meta["next_free_vgpr"]) <= 128                      # 4 waves per SIMD
    # the step loop of the plain kernel: no fused multiply-add between the two waits' worth of code
    lines = body.splitlines()
    waits = [i for i, l in enumerate(lines) if "CBET_RECORD_WAIT" in l]
    issues = [i for i, l in enumerate(lines) if "CBET_RECORD_ISSUE" in l]
    hot = lines[issues[1] - 150:waits[1] + 200]                    # move / relocate / gather / flush / weights / wait / sums
    fused = [l.split(";")[0].split() for l in hot if "v_fma_f64" in l or "v_fmac_f64" in l]
    # ... except the eight the source asks for by name: the pending sums, S = fma(S, k, w) with k = 1.0 or 0.0 (exact
    # forms of S + w and w), all with the same multiplier register
    # (v_fma_f64 S, S, k, w: written in place from inline assembly)
    # ... once in the gather's shadow (the previous step's deposit) and once more where a ray ends (its last deposit joins the
    # sums before they are handed over): groups of eight
    assert len(fused) in (8, 16) and all(f[0] == "v_fma_f64" and f[1] == f[2] for f in fused), fused
    for g in range(0, len(fused), 8):
        assert len({f[3] for f in fused[g:g + 8]}) == 1, fused[g:g + 8]


# ---- gfx940 / gfx950 data hazards the compiler does not track into inline assembly ---------------------------------
# A VALU instruction that WRITES an SGPR (v_readlane / v_readfirstlane reloading a spilled scalar, a v_cmp into a scalar
# pair) must be followed by 5 wait states before a vector-memory instruction reads that SGPR and by 2 before a VALU
# instruction reads it as a constant.  Inside ;;#ASMSTART ... ;;#ASMEND the compiler inserts nothing, so the fragment
# itself has to carry the s_nop (s_nop N = N + 1 wait states; any other instruction = 1) unless the straight-line code in
# front of it provably holds no such write that close.
_VALU_SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32", "v_cmp", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co",
                      "v_subbrev_co", "v_subrev_co", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")


def _sgprs(text):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    if re.search(r"\bvcc\b", text):
        out.add(-1)
    return out


def _code(line):
    return line.split(";")[0].strip()


def _states(code):
    m = re.match(r"s_nop\s+(\d+)", code)
    return int(m.group(1)) + 1 if m else 1


def _writes_sgpr(code, regs):
    op = code.split()[0]
    if not op.startswith(_VALU_SGPR_WRITERS):
        return False
    dst = code[len(op):].split(",")
    # destination operands: the first (and for carry-out forms the second) operand
    first = ",".join(dst[:2]) if ("_co_" in op or op.startswith(("v_mad_u64", "v_mad_i64", "v_div_scale"))) else dst[0]
    if op.startswith("v_cmp") and op.endswith("_e32"):
        first = "vcc"
    return bool(_sgprs(first) & regs)


def _scalar_sources(code):
    """SGPRs an instruction of an assembly fragment READS (everything but the destination of a v_/global_ instruction)."""
    op = code.split()[0]
    ops = code[len(op):].split(",")
    return _sgprs(",".join(ops[1:])) if op.startswith(("v_", "global_load", "buffer_load", "flat_load", "ds_read")) else _sgprs(",".join(ops))


def hazard_violations(body):
    lines = body.splitlines()
    bad = []
    i = 0
    while i < len(lines):
        if ";;#ASMSTART" not in lines[i]:
            i += 1
            continue
        start = i
        states = 0
        i += 1
        while i < len(lines) and ";;#ASMEND" not in lines[i]:
            code = _code(lines[i])
            if code and not code.endswith(":"):
                op = code.split()[0]
                need = 5 if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else (2 if op.startswith("v_") else 0)
                src = _scalar_sources(code) if need else set()
                if src and states < need:
                    # not covered inside the fragment: the code in front of it must hold no VALU write of these registers
                    # within the missing wait states; a label or a branch target in that window proves nothing
                    missing, j = need - states, start - 1
                    while missing > 0 and j >= 0:
                        c = _code(lines[j])
                        j -= 1
                        if not c or c.startswith((";;#", ".")) and not c.endswith(":"):
                            continue
                        if c.endswith(":") or c.split()[0].startswith(("s_cbranch", "s_branch", "s_endpgm")):
                            bad.append((start + 1, code, "control flow inside the hazard window"))
                            break
                        if _writes_sgpr(c, src):
                            bad.append((start + 1, code, "needs %d wait states after: %s" % (need, c)))
                            break
                        missing -= _states(c)
                states += _states(code)
            i += 1
        i += 1
    return bad


def test_hazard_walker_sees_a_missing_nop():
    """The walker itself: the diet1 fault of round 4 (v_readlane reloads the table base, the gather reads it at once)."""
    faulty = """
	v_readlane_b32 s24, v126, 7
	v_readlane_b32 s25, v126, 8
	;;#ASMSTART
	global_load_dwordx4 v[2:5], v53, s[24:25]
	;;#ASMEND
"""
    fixed = faulty.replace(";;#ASMSTART\n", ";;#ASMSTART\n\ts_nop 4\n")
    mad = """
	v_readlane_b32 s65, v126, 2
	;;#ASMSTART
	v_mad_i32_i24 v2, v76, s65, v18
	;;#ASMEND
"""
    assert hazard_violations(faulty) and not hazard_violations(fixed)
    assert hazard_violations(mad) and not hazard_violations(mad.replace(";;#ASMSTART\n", ";;#ASMSTART\n\ts_nop 1\n"))
    far = "\tv_readlane_b32 s65, v126, 2\n\tv_mov_b32_e32 v1, v2\n\tv_mov_b32_e32 v3, v2\n" + mad.split("\n", 2)[2]
    assert not hazard_violations(far)


def test_inline_assembly_scalar_reads_have_their_wait_states(listing):
    """All sixteen instantiations: every inline-assembly instruction with an SGPR source is far enough behind the last VALU
    write of that SGPR (5 wait states for vector memory, 2 for VALU)."""
    for name, body in listing.items():
        bad = hazard_violations(body)
        assert not bad, (name, bad[:5])
