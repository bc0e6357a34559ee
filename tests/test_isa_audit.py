"""The shipped kernel's hand-scheduled pieces, checked in the gfx950 assembly (cross-compiled here, no GPU needed).

cbet_trace_window.hip issues the per-step record gather from inline assembly and waits for it with a counted
`s_waitcnt vmcnt(N)`, so the compiler does not know that the destination registers are written asynchronously
between the two.  That is only correct if it leaves the record where the load puts it: both assembly blocks print
the registers they were given (CBET_RECORD_ISSUE / CBET_RECORD_WAIT comments) and this test requires, for every
instantiation of the kernel, that they all name the same registers and that nothing in between touches them.
It also pins the properties DESIGN.md quotes: no fused multiply-add in the kernels (one IEEE operation per reference
statement), native fp64 atomics (no compare-and-swap loops), no scratch, the LDS size that gives 14 waves per CU."""
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
    for inst in ("ILi16ELb0ELi0E", "ILi16ELb1ELi0E", "ILi16ELb0ELi1E", "ILi16ELb1ELi1E", "ILi16ELb0ELi2E", "ILi16ELb1ELi2E",
                 "ILi8ELb0ELi4E", "ILi8ELb1ELi4E"):
        assert inst in names, inst
    assert len(listing) == 8


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
    counted in WaveCounters::pend: every global_add call site of the shipped kernel sits behind a `wc.pend +=` of its
    own (the four-component field pass waits with vmcnt(0) and is exempt).  The file refuses targets other than gfx950
    (one in-order vmcnt shared by loads, stores and atomics)."""
    src = open(os.path.join(CSRC, "cbet_trace_window.hip")).read().splitlines()
    sites = [i for i, l in enumerate(src) if "global_add(a," in l and "__device__" not in l]
    assert len(sites) >= 12
    for i in sites:
        window = "\n".join(src[max(0, i - 12):i + 1])
        assert "wc.pend +=" in window or "comp_stride + own_node" in window, (i + 1, src[i])
    text = "\n".join(src)
    assert "#error" in text and "__gfx950__" in text


def test_native_fp64_atomics(listing):
    """HBM and LDS adds are the native instructions, never compare-and-swap loops or flat atomics."""
    for name, body in listing.items():
        assert "global_atomic_cmpswap" not in body and "flat_atomic" not in body and "ds_cmpst" not in body, name
        assert "global_atomic_add_f64" in body and "ds_add_f64" in body, name


def test_resources_of_the_headline_instance(listing):
    (name, body), = [(n, b) for n, b in listing.items() if "ILi16ELb0ELi0E" in n]
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
    assert len(fused) == 8 and all(f[0] == "v_fma_f64" and f[1] == f[2] for f in fused) and len({f[3] for f in fused}) == 1, fused
