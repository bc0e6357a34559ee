"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerance: BASELINE.json asks for <= 1e-4 max rel-err (metric of SURVEY.md 8(c), conftest.parity_err).
The kernels are built with -ffp-contract=off so each ray's arithmetic is the oracle's operation
sequence; what remains is fp64 summation order, so the tests hold the path to PARITY_TOL = 1e-9
(observed ~1e-13) -- five orders tighter than the stated bound -- and to EXACT ray-step counts.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, NCPU, parity_err

pytestmark = pytest.mark.gpu

PARITY_TOL = 1e-9          # asserted;  BASELINE.json's bound is 1e-4
VARIANTS = [1, 2, 3]     # GLOBAL_ATOMICS, LDS_COMBINE (tagged), LDS_WINDOW (default)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "the gpu tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def api():
    from cbet_raytracing_3d_amd import api as a
    a.lib()   # raises if the HIP library was not built -- no fallback
    return a


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "golden.json")))


def make_tracer(api, inputs, n, **kw):
    from cbet_raytracing_3d_amd.tracer import RayTracer
    bn, r, ne, te = inputs
    beams = kw.pop("beams", None)
    ne = kw.pop("ne", ne)
    te = kw.pop("te", te)
    p = api.default_params(n, **kw)
    if beams is not None:
        bn = bn[beams]
        p.nbeams = len(beams)
    else:
        bn = bn[:p.nbeams]
    return RayTracer(p, r, ne, te, beam_norm=bn)


def run(tr, torch, **kw):
    e = tr.new_grid()
    tr.counters(reset=True)
    tr.launch(e, **kw)
    c = tr.counters(reset=True)
    return e.cpu().numpy(), c


def test_helpers_roundtrip_and_errors(api, torch_cuda):
    """multi_gpu.cpp:3-28, 44-59 semantics through the ABI."""
    host = np.arange(1000, dtype=np.float64)
    back = np.zeros_like(host)
    d = api.safeGPUAlloc(host.nbytes, 0)
    assert d
    api.moveToAndFromGPU(d, host, host.nbytes, 0)      # H2D, direction inferred
    d2 = api.safeGPUAlloc(host.nbytes, 0)
    api.moveToAndFromGPU(d2, d, host.nbytes, 0)        # D2D
    api.moveToAndFromGPU(back, d2, host.nbytes, 0)     # D2H
    assert np.array_equal(back, host)
    api.gpuFree(d, 0)
    api.gpuFree(d2, 0)
    with pytest.raises(api.CbetError) as ei:           # free < size guard
        api.safeGPUAlloc(1 << 50, 0)
    assert ei.value.code == api.ENOMEM
    with pytest.raises(api.CbetError) as ei:
        api.moveToAndFromGPU(back, host, 8, -1)
    assert ei.value.code == api.ENODEVICE
    with pytest.raises(api.CbetError) as ei:
        api.safeGPUAlloc(8, 999)
    assert ei.value.code == api.ENODEVICE


def test_node_tables_match_oracle(api, oracle, inputs, torch_cuda):
    bn, r, ne, te = inputs
    tr = make_tracer(api, inputs, 48)
    e = tr.new_grid()
    tr.launch(e)
    torch_cuda.cuda.synchronize()
    ne3d, kap = tr.node_tables()
    one3d, okap = oracle.node_tables(oracle.default_config(48), r, ne, te)
    assert np.abs(ne3d / one3d - 1).max() < 1e-15 and np.abs(kap / okap - 1).max() < 1e-14
    exact = (ne3d == one3d).mean(), (kap == okap).mean()
    print("node tables bitwise-equal fraction ne3d=%.6f kappa=%.6f" % exact)
    tr.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_config1_two_beams_uniform_plasma(api, oracle, inputs, torch_cuda, golden, variant):
    """BASELINE config 1: 2 crossing beams, 64^3 uniform plasma, refraction only, vs the serial CPU loop."""
    bn, r, _, _ = inputs
    g = golden["config1_64"]
    d = api.derive(api.default_params(64))
    ne_u, te_u = np.full(443, g["ne_over_ncrit"] * d.ncrit), np.full(443, g["te"])
    tr = make_tracer(api, inputs, 64, beams=g["beams"], absorption=0, ne=ne_u, te=te_u)
    e, c = run(tr, torch_cuda, kernel_variant=variant)
    assert c.ray_steps == g["ray_steps"]
    cfg = oracle.default_config(64, nbeams=2, absorption=0)
    oe, osteps = oracle.trace(cfg, bn[g["beams"]].copy(), r, ne_u, te_u, nthreads=1)   # serial ray loop
    assert osteps == c.ray_steps
    assert parity_err(e, oe) < PARITY_TOL
    assert np.count_nonzero(e) == g["nonzero"]
    planes = np.load(os.path.join(GOLDEN, "planes_cfg1_64.npz"))
    assert parity_err(e[33], planes["yz"]) < PARITY_TOL
    tr.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_full_grid_parity_64(api, oracle, inputs, torch_cuda, golden, variant):
    """60 beams, s83177, 64^3: every one of the 66^3 cells against the oracle."""
    bn, r, ne, te = inputs
    tr = make_tracer(api, inputs, 64)
    e, c = run(tr, torch_cuda, kernel_variant=variant)
    oe, osteps, per_beam = oracle.trace(oracle.default_config(64), bn, r, ne, te, nthreads=NCPU,
                                        want_per_beam=True)
    g = golden["cases"][0]
    assert c.ray_steps == osteps == g["ray_steps"] == 30712072
    assert c.rays_traced == 60 * tr.derived.nlive_rays
    err = parity_err(e, oe)
    print("64^3 variant %d: parity err %.3e, global atomics per step %.3f" %
          (variant, err, c.global_atomics / c.ray_steps))
    assert err < PARITY_TOL
    assert np.array_equal(e == 0, oe == 0)              # the over-critical core stays exactly 0
    assert np.array_equal(e < 0, oe < 0)
    tr.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_truth_100_golden(api, oracle, inputs, torch_cuda, golden, variant, tmp_path):
    """BASELINE config 2 (the reference's `make test` case, Makefile:14-17): golden planes, scalar known answers and
    the 6-digit text dump byte-identical to the one whose digest the survey's host compile of the reference kernel
    recorded (the reference's golden file truth_100 is absent from the mount; nobody has compared with it)."""
    tr = make_tracer(api, inputs, 100)
    e, c = run(tr, torch_cuda, kernel_variant=variant)
    g = golden["cases"][1]
    assert c.ray_steps == g["ray_steps"] == 124789870
    assert c.rays_traced == 60 * 15102
    assert np.count_nonzero(e) == g["nonzero"] == 1055570
    assert int((e < 0).sum()) == 1 and e[101, 7, 0] == pytest.approx(-7512.4896951, rel=1e-10)
    assert e.sum() == pytest.approx(g["sum"], rel=1e-12)
    assert e.max() == pytest.approx(g["max"], rel=1e-12)
    assert e[1, 1, 1] == pytest.approx(668336.05188607727, rel=1e-12)
    assert e[51, 51, 90] == pytest.approx(23295164142.968483, rel=1e-12)
    assert e[51, 51, 51] == 0.0
    planes = np.load(os.path.join(GOLDEN, "planes_100.npz"))
    for name, got in (("yz", e[51]), ("xz", e[:, 51]), ("xy", e[:, :, 51]), ("face_x0", e[0]),
                      ("face_z0", e[:, :, 0])):
        assert parity_err(got, planes[name]) < PARITY_TOL, name
    path = str(tmp_path / "edep.txt")
    assert api.write_text(e, path) == g["text_bytes"] == 12544620        # the product's own writer
    assert hashlib.md5(open(path, "rb").read()).hexdigest() == g["text_md5"] == "cc0909ed1c5938704c51165dc20cb829"
    assert oracle.write_text(e, path) == g["text_bytes"]
    assert hashlib.md5(open(path, "rb").read()).hexdigest() == g["text_md5"]
    tr.close()


def test_sharding_and_beam_independence(api, inputs, torch_cuda):
    """edep(all) == sum over shards == sum over beams, to summation-order rounding."""
    tr = make_tracer(api, inputs, 64)
    full, c = run(tr, torch_cuda)
    parts = tr.new_grid()
    steps = 0
    for s in range(3):
        tr.launch(parts, shard_index=s, shard_count=3)
        steps += tr.counters(reset=True).ray_steps
    assert steps == c.ray_steps
    assert parity_err(parts.cpu().numpy(), full) < 1e-11
    by_beam = tr.new_grid()
    for lo in range(0, 60, 20):
        tr.launch(by_beam, beam_lo=lo, beam_hi=lo + 20)
    assert parity_err(by_beam.cpu().numpy(), full) < 1e-11
    # the reference's own split rule: b-th block of nbeams/ngpus beams (launch_ray_XZ.cu:123)
    assert tr.params.beam_hi == -1   # CBET_BEAMS_BY_GPU: cbet_params_default leaves the range unset
    p = tr.params.copy(ngpus=2, beam_lo=0, beam_hi=-1)
    halves = tr.new_grid()
    d = tr.derived
    stream = torch_cuda.cuda.current_stream().cuda_stream
    for b in range(2):
        api.launch_ray_XYZ(b, d.nindices, tr.d_te, tr.d_r, tr.d_ne, halves, tr.d_bbeam_norm,
                           tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, d.xconst, d.yconst, d.zconst,
                           p, ctx=tr.ctx, stream=stream)
    assert parity_err(halves.cpu().numpy(), full) < 1e-11
    # an explicit empty range -- [0,0) is rank 0's share when there are more ranks than beams -- traces nothing
    tr.counters(reset=True)
    nothing = tr.new_grid()
    for lo in (0, 7, 60):
        api.launch_ray_XYZ(0, d.nindices, tr.d_te, tr.d_r, tr.d_ne, nothing, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r,
                           tr.d_phase_r, d.xconst, d.yconst, d.zconst, tr.params.copy(ngpus=2, beam_lo=lo, beam_hi=lo),
                           ctx=tr.ctx, stream=stream)
    torch_cuda.cuda.synchronize()
    assert tr.counters().ray_steps == 0 and float(nothing.abs().max()) == 0.0
    tr.close()


def test_in_place_edit_of_context_tables_rebuilds_step_records(api, inputs, torch_cuda):
    """cbet_context_tables hands out writable pointers: a launch that uses the context's own tables after an
    in-place edit announced by that call must gather from rebuilt step records, not from the cached ones."""
    tr = make_tracer(api, inputs, 48, nbeams=6)
    ref, c = run(tr, torch_cuda)                       # tabulates the context's tables and caches their records
    d = tr.derived
    n3 = 48 ** 3
    stream = torch_cuda.cuda.current_stream().cuda_stream
    p = tr.params.copy(beam_lo=0, beam_hi=6)

    def trace(ne3d, kap):
        e = tr.new_grid()
        tr.counters(reset=True)
        api.trace_nodes(0, d.nindices, ne3d, kap, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                        d.xconst, d.yconst, d.zconst, p, tr.ctx, stream)
        return e.cpu().numpy(), tr.counters(reset=True)

    same, c1 = trace(None, None)                       # untouched tables: the cached records are right
    assert c1.ray_steps == c.ray_steps and parity_err(same, ref) < 1e-11
    ne_addr, kap_addr = tr.ctx.tables()                # announces the edit below
    host = np.empty(n3)
    api.moveToAndFromGPU(host, kap_addr, 8 * n3, tr.gpu)
    host[: n3 // 2] *= 0.5                             # weaker absorption in the x < 0 half-space
    api.moveToAndFromGPU(kap_addr, host, 8 * n3, tr.gpu)
    edited, c2 = trace(None, None)
    ne_h = np.empty(n3)
    api.moveToAndFromGPU(ne_h, ne_addr, 8 * n3, tr.gpu)
    want, c3 = trace(torch_cuda.from_numpy(ne_h).cuda(), torch_cuda.from_numpy(host).cuda())   # caller-owned copies
    assert c2.ray_steps == c3.ray_steps > c.ray_steps
    assert parity_err(edited, want) < 1e-11
    tr.close()


def test_accumulates_into_edep(api, inputs, torch_cuda):
    """edep is added into, never cleared (launch_ray_XZ.cu:341-348)."""
    tr = make_tracer(api, inputs, 64, nbeams=6)
    a, _ = run(tr, torch_cuda, kernel_variant=2)
    e = tr.new_grid()
    tr.launch(e)
    tr.launch(e)
    assert parity_err(e.cpu().numpy(), 2 * a) < 1e-11
    tr.close()


def test_device_trig_fallback_is_close(api, inputs, torch_cuda):
    """bbeam_norm == NULL evaluates acos/atan2/cos/sin on the device (launch_ray_XZ.cu:99-111 as
    written); device libm may differ from the host's in the last bit, hence 1e-4, not 1e-9."""
    tr = make_tracer(api, inputs, 64, nbeams=8)
    a, ca = run(tr, torch_cuda)
    b, cb = run(tr, torch_cuda, use_host_trig=False)
    assert abs(int(ca.ray_steps) - int(cb.ray_steps)) <= 1e-4 * ca.ray_steps
    assert abs(b.sum() / a.sum() - 1) < 1e-6
    tr.close()


def test_invalid_launches_fail_loudly(api, inputs, torch_cuda):
    tr = make_tracer(api, inputs, 32, nbeams=2)
    e = tr.new_grid()
    d = tr.derived
    args = (tr.d_te, tr.d_r, tr.d_ne, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
            d.xconst, d.yconst, d.zconst)
    with pytest.raises(api.CbetError) as ei:                       # geometry mismatch with workspace
        api.launch_ray_XYZ(0, d.nindices, *args, api.default_params(48, nbeams=2), ctx=tr.ctx)
    assert ei.value.code == api.EINVAL
    with pytest.raises(api.CbetError):                             # wrong nindices
        api.launch_ray_XYZ(0, d.nindices + 1, *args, tr.params, ctx=tr.ctx)
    with pytest.raises(api.CbetError):                             # beam range outside the table
        api.launch_ray_XYZ(0, d.nindices, *args, tr.params.copy(beam_lo=1, beam_hi=3), ctx=tr.ctx)
    with pytest.raises(api.CbetError):
        api.launch_ray_XYZ(0, d.nindices, *args, tr.params.copy(kernel_variant=77), ctx=tr.ctx)
    api.launch_ray_XYZ(0, 0, *args, tr.params, ctx=tr.ctx)         # nindices = 0: ray loop skipped
    torch_cuda.cuda.synchronize()
    assert float(e.abs().sum()) == 0.0
    api.launch_ray_XYZ(0, d.nindices, *args, tr.params, ctx=None)  # default per-device workspace
    torch_cuda.cuda.synchronize()
    assert float(e.sum()) > 0
    tr.close()


def test_ray_tracing_orchestrator(api, oracle, inputs, torch_cuda):
    """cbet_ray_tracing (rayTracing(), main.cu:96-232) on one device: host arrays in, edep += out."""
    bn, r, ne, te = inputs
    p = api.default_params(48, nbeams=10)
    edep = np.full((50, 50, 50), 1.0)                   # pre-existing content must be kept (+=)
    timers, cnt = api.ray_tracing(te, r, ne, edep, p, beam_norm=bn[:10])
    oe, osteps = oracle.trace(oracle.default_config(48, nbeams=10), bn[:10].copy(), r, ne, te, nthreads=NCPU)
    assert cnt.ray_steps == osteps
    assert parity_err(edep - 1.0, oe) < 1e-9
    assert timers["total"] >= timers["tracing"] > 0


def test_nonuniform_grid_and_rays_per_zone(api, oracle, inputs, torch_cuda):
    """Ragged case: nx != ny != nz and rays_per_zone = 3 (patches no longer align with zones)."""
    bn, r, ne, te = inputs
    from cbet_raytracing_3d_amd.tracer import RayTracer
    p = api.default_params(40, nbeams=4, rays_per_zone=3)
    p.ny, p.nz = 48, 36
    tr = RayTracer(p, r, ne, te, beam_norm=bn[[1, 12, 33, 58]])
    e, c = run(tr, torch_cuda)
    cfg = oracle.default_config(40, nbeams=4, rays_per_zone=3)
    cfg.ny, cfg.nz = 48, 36
    oe, osteps = oracle.trace(cfg, bn[[1, 12, 33, 58]].copy(), r, ne, te, nthreads=NCPU)
    assert c.ray_steps == osteps
    assert parity_err(e, oe) < PARITY_TOL
    tr.close()


def test_full_size_256_properties(api, inputs, torch_cuda):
    """BASELINE config 3 size.  The oracle needs minutes here, so check the size-independent facts:
    SURVEY.md 8(c) known answers at 256^3, energy bookkeeping and variant agreement."""
    tr = make_tracer(api, inputs, 256)
    e2, c2 = run(tr, torch_cuda, kernel_variant=2)
    assert c2.ray_steps == 2123497670
    assert c2.rays_traced == 60 * 98872
    assert e2.sum() == pytest.approx(1.0076068555e19, rel=1e-10)
    assert e2.max() == pytest.approx(4.0037106759e13, rel=1e-10)
    assert np.count_nonzero(e2) == 17053618 and e2.size == 17173512
    e1, c1 = run(tr, torch_cuda, kernel_variant=1)
    assert c1.ray_steps == c2.ray_steps and c1.global_atomics == 8 * c1.ray_steps
    assert parity_err(e2, e1) < 1e-10
    print("256^3: LDS-combine global atomics per ray-step = %.3f (global variant: 8)" %
          (c2.global_atomics / c2.ray_steps))
    tr.close()


def test_full_grid_parity_256_default_kernel(api, oracle, inputs, torch_cuda):
    """BASELINE config 3 with the kernel bench.py times (kernel_variant = 3, every knob on auto): all
    258^3 cells against the oracle run on this box's host cores (~1 min), the reference's whole-grid
    criterion (Makefile:14-17) at the headline size.  Exact ray-step count, sum, max, zero pattern."""
    bn, r, ne, te = inputs
    tr = make_tracer(api, inputs, 256)
    e, c = run(tr, torch_cuda, kernel_variant=3)
    assert c.ray_steps == 2123497670 and c.rays_traced == 60 * 98872
    cfg = oracle.default_config(256)
    oe, osteps = np.zeros(oracle.grid_shape(cfg)), 0
    for lo in range(0, 60, 10):
        _, st = oracle.trace(cfg, bn.copy(), r, ne, te, beam_lo=lo, beam_hi=lo + 10, nthreads=NCPU, edep=oe)
        osteps += st
        print("  oracle beams %d-%d done" % (lo, lo + 9), flush=True)
    assert osteps == c.ray_steps
    assert parity_err(e, oe) < PARITY_TOL
    assert e.sum() == pytest.approx(1.0076068555e19, rel=1e-10)
    assert e.max() == pytest.approx(4.0037106759e13, rel=1e-10)
    assert np.count_nonzero(e) == 17053618 and e.size == 17173512
    assert np.array_equal(e == 0, oe == 0)                                       # the over-critical core stays exactly 0
    e0, c0 = run(tr, torch_cuda, stats=True)                                     # kernel_variant = 0 (auto) is the same kernel
    assert c0.ray_steps == c.ray_steps and 0 < c0.global_atomics < c.ray_steps
    assert c.global_atomics == 0 and c.wave_steps == 0                           # (window diagnostics only on request)
    assert parity_err(e0, oe) < PARITY_TOL
    tr.close()


def test_config5_as_stated_512_six_rays_per_zone(api, inputs, torch_cuda):
    """BASELINE config 5 as stated: 512^3, 60 beams x 1.13e6 ray ids per beam (rays_per_zone = 6; def.cuh:58
    ships 4), one pass.  No oracle at this size in test time, so the size-independent facts: every live ray
    launched, the over-critical core exactly zero, deposited energy equal to the 4-rays/zone pass within the
    discretisation (uray_mult carries 1/rpz^2, def.cuh:92), the default kernel and the tagged LDS scheme
    agreeing cell by cell with equal step counts."""
    tr = make_tracer(api, inputs, 512, rays_per_zone=6)
    d = tr.derived
    assert (d.nrays_x, d.nrays, d.nt) == (1062, 1062 * 1062, 2048)
    e3, c3 = run(tr, torch_cuda, kernel_variant=3, stats=True)
    assert c3.rays_traced == 60 * d.nlive_rays and d.nlive_rays == 883790
    assert 3.5e10 < c3.ray_steps < 4.1e10
    assert e3[257, 257, 257] == 0.0 and (e3[250:264, 250:264, 250:264] == 0).all()
    total3 = float(e3.sum())
    assert 0.9 < total3 / (1.0076068555e19 * 4.0) < 1.1          # sum(edep) ~ n^2, independent of rays/zone
    e2, c2 = run(tr, torch_cuda, kernel_variant=2)
    assert c2.ray_steps == c3.ray_steps
    assert parity_err(e3, e2) < 1e-10
    print("512^3, 6 rays/zone: %d ray-steps, %.3f global atomics/step" % (c3.ray_steps, c3.global_atomics / c3.ray_steps))
    tr.close()


def test_cli_driver_reproduces_truth_100():
    """tools/cbet_gpu.cpp = main.cu's driver over the C ABI: `cbet-gpu 10 --print` must emit the
    reference's golden text (Makefile:14-17), and without --print the four timers (main.cu:225-230)."""
    import subprocess
    from cbet_raytracing_3d_amd import build
    from conftest import ROOT
    exe = build.CLI_PATH
    assert os.path.exists(exe), "cbet-gpu not built"
    out = subprocess.run([exe, "10", "--n", "100", "--print"], cwd=ROOT, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    assert len(out.stdout) == 12544620
    assert hashlib.md5(out.stdout).hexdigest() == "cc0909ed1c5938704c51165dc20cb829"
    out = subprocess.run([exe, "10", "--n", "64"], cwd=ROOT, capture_output=True, timeout=300, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].startswith("rt: Init ") and lines[1].startswith("Tracing ")
    assert lines[2].startswith("Combining ") and lines[3].startswith("Total ")
    assert "ray-steps 30712072" in lines[4]


def test_trace_with_caller_supplied_node_tables(api, oracle, inputs, torch_cuda):
    """the 3-D plasma entry.  Tables supplied by the caller (here: the oracle's
    node tables, uploaded) must give the same grid as the radial-profile launch; a non-spherical
    edit of kappa3d changes only the absorption, energy still balances."""
    bn, r, ne, te = inputs
    tr = make_tracer(api, inputs, 48, nbeams=6)
    ref, c = run(tr, torch_cuda)
    one3d, okap = oracle.node_tables(oracle.default_config(48), r, ne, te)
    d_ne = torch_cuda.from_numpy(one3d).cuda()
    d_kap = torch_cuda.from_numpy(okap).cuda()
    d = tr.derived
    stream = torch_cuda.cuda.current_stream().cuda_stream

    def trace(kap):
        e = tr.new_grid()
        tr.counters(reset=True)
        api.trace_nodes(0, d.nindices, d_ne, kap, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r,
                        tr.d_phase_r, d.xconst, d.yconst, d.zconst, tr.params.copy(beam_lo=0, beam_hi=6),
                        tr.ctx, stream)
        return e.cpu().numpy(), tr.counters(reset=True)

    got, c2 = trace(d_kap)
    assert c2.ray_steps == c.ray_steps and parity_err(got, ref) < 1e-11
    half = d_kap.clone()
    half[:24] *= 0.5                                   # weaker absorption in the x < 0 half-space
    got2, c3 = trace(half)
    assert c3.ray_steps > c.ray_steps                  # rays survive longer
    # beams 0-5 enter from +x (omega_beams.h rows 0-29 have n_x >= 0): upstream half untouched,
    # downstream half absorbs differently
    assert parity_err(got2[30:], got[30:]) < 1e-11
    assert abs(got2[:22].sum() / got[:22].sum() - 1) > 1e-3
    tr.close()


def test_patch_orders_trace_the_same_rays(api, inputs, oracle, torch_cuda):
    """cbet_params.patch_order only changes which bundle a workgroup takes: Morton (0), longest first (1), k radial
    rings with Morton inside (k >= 2) -- every order lists every live ray once and deposits the oracle's grid."""
    bn, r, ne, te = inputs
    cfg = oracle.default_config(40, nbeams=3)
    want, steps = oracle.trace(cfg, bn[[0, 17, 42]], r, ne, te, nthreads=NCPU)
    lists = []
    for order in (0, 1, 2, 5, 64):
        tr = make_tracer(api, inputs, 40, beams=[0, 17, 42], patch_order=order)
        live = api.live_ray_list(tr.params)
        lists.append(np.sort(live[live >= 0]))
        assert len(live) % 64 == 0
        e, c = run(tr, torch_cuda)
        assert c.ray_steps == steps, order
        assert parity_err(e, want) < PARITY_TOL, order
        tr.close()
    for l in lists[1:]:
        assert np.array_equal(l, lists[0])


def test_rim_merge_traces_the_same_rays(api, inputs, oracle, torch_cuda):
    """cbet_params.rim_merge packs neighbouring rim patches of the beam into one bundle (default: a footprint of 4
    launch zones = 16 rays): off, the default and wider footprints list every live ray once, deposit the oracle's grid with the
    oracle's step count -- in all three kernel formulations -- and the packed lists are shorter."""
    bn, r, ne, te = inputs
    beams = [3, 21, 40]
    cfg = oracle.default_config(72, nbeams=3)
    want, steps = oracle.trace(cfg, bn[beams], r, ne, te, nthreads=NCPU)
    lists, lengths = [], []
    for merge in (0, 4, 6, 16):
        tr = make_tracer(api, inputs, 72, beams=beams, rim_merge=merge)
        live = api.live_ray_list(tr.params)
        lists.append(np.sort(live[live >= 0]))
        lengths.append(len(live))
        for variant in ((3, 1, 2) if merge in (0, 4) else (3,)):
            e, c = run(tr, torch_cuda, kernel_variant=variant)
            assert c.ray_steps == steps, (merge, variant)
            assert parity_err(e, want) < PARITY_TOL, (merge, variant)
        tr.close()
    assert all(np.array_equal(l, lists[0]) for l in lists[1:])
    assert lengths[0] > lengths[1] >= lengths[2] >= lengths[3]
    tr = make_tracer(api, inputs, 40, beams=[0])
    d = tr.derived
    with pytest.raises(api.CbetError) as ei:     # the packing is part of the geometry a context is created for
        api.launch_ray_XYZ(0, d.nindices, tr.d_te, tr.d_r, tr.d_ne, tr.new_grid(), tr.d_bbeam_norm, tr.d_beam_norm,
                           tr.d_pow_r, tr.d_phase_r, d.xconst, d.yconst, d.zconst, tr.params.copy(rim_merge=0), ctx=tr.ctx)
    assert ei.value.code == api.EINVAL
    tr.close()


def test_a_callers_launch_list_regroups_the_same_rays(api, inputs, oracle, torch_cuda):
    """cbet_context_set_launch_list: any regrouping of the context's live rays into bundles of 64 traces the same
    rays (here: bundles reversed, lanes rotated, and the rays dealt round-robin into twice as many half-empty bundles);
    a list that drops, repeats or invents a ray, or has an empty bundle, is refused."""
    bn, r, ne, te = inputs
    tr = make_tracer(api, inputs, 48, beams=[5, 33])
    want, steps = oracle.trace(oracle.default_config(48, nbeams=2), bn[[5, 33]], r, ne, te, nthreads=NCPU)
    live = api.live_ray_list(tr.params).reshape(-1, 64)
    e0, c0 = run(tr, torch_cuda)
    assert c0.ray_steps == steps and parity_err(e0, want) < PARITY_TOL
    ids = live[live >= 0]
    dealt = -np.ones((2 * len(live), 64), dtype=np.int32)
    for k, v in enumerate(ids):
        dealt[k % len(dealt), k // len(dealt)] = v
    dealt = dealt[(dealt >= 0).any(1)]
    for lists in (np.roll(live[::-1], 5, axis=1), dealt):
        tr.ctx.set_launch_list(lists.ravel())
        e, c = run(tr, torch_cuda)
        assert c.ray_steps == steps
        assert parity_err(e, want) < PARITY_TOL
    bad = live.copy().ravel()
    first = int(np.nonzero(bad >= 0)[0][0])
    for wrong in (bad[:-64], np.concatenate([bad, -np.ones(64, dtype=bad.dtype)]), bad[:-1]):
        with pytest.raises(api.CbetError) as ei:
            tr.ctx.set_launch_list(wrong)
        assert ei.value.code == api.EINVAL
    dup = bad.copy(); dup[first] = bad[np.nonzero(bad >= 0)[0][1]]
    inv = bad.copy(); inv[first] = -7
    for wrong in (dup, inv):
        with pytest.raises(api.CbetError) as ei:
            tr.ctx.set_launch_list(wrong)
        assert ei.value.code == api.EINVAL
    e, c = run(tr, torch_cuda)                      # the refused calls left the last good list in place
    assert c.ray_steps == steps and parity_err(e, want) < PARITY_TOL
    tr.close()


def test_window_kernel_combines_and_is_parity_exact(api, oracle, inputs, torch_cuda):
    """The LDS windows only reorder fp64 sums: same grid and step count as the oracle on a beam subset whose
    bundles fan out in every direction, with most deposits combined in LDS before they reach HBM."""
    bn, r, ne, te = inputs
    beams = [0, 7, 19, 23, 31, 38, 44, 52, 57, 59]
    tr = make_tracer(api, inputs, 56, beams=beams)
    e, c = run(tr, torch_cuda, kernel_variant=3, stats=True)     # (cbet_params.window_stats: the counting instantiation)
    oe, osteps = oracle.trace(oracle.default_config(56, nbeams=len(beams)), bn[beams].copy(), r, ne, te,
                              nthreads=NCPU)
    assert c.ray_steps == osteps
    assert parity_err(e, oe) < PARITY_TOL
    e_plain, c_plain = run(tr, torch_cuda, kernel_variant=3)     # ... and the one that does not count: same rays, same grid
    assert c_plain.ray_steps == osteps and c_plain.rays_traced == c.rays_traced and parity_err(e_plain, oe) < PARITY_TOL
    assert 0 < c.global_atomics < 0.6 * c.ray_steps      # the window really combines
    assert 64 * c.wave_steps >= c.ray_steps > 32 * c.wave_steps
    assert c.wave_steps_miss < 0.2 * c.wave_steps and c.lds_evictions < 0.05 * c.ray_steps
    tr.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_wide_index_path(api, oracle, inputs, torch_cuda, variant):
    """Grids with 8*nx*ny*nz >= 2^32 bytes (n > 812) switch to 64-bit node-table indexing; force
    that code path on a small grid and hold it to the same parity."""
    bn, r, ne, te = inputs
    tr = make_tracer(api, inputs, 48, nbeams=8)
    e, c = run(tr, torch_cuda, kernel_variant=variant, force_wide_index=1)
    oe, osteps = oracle.trace(oracle.default_config(48, nbeams=8), bn[:8].copy(), r, ne, te, nthreads=NCPU)
    assert c.ray_steps == osteps and parity_err(e, oe) < PARITY_TOL
    tr.close()


def test_stress_512_properties(api, inputs, torch_cuda):
    """BASELINE config 5 size (512^3, 60 beams; 30.1 M ray ids, edep 1.09 GB).  No oracle at this
    size in test time; check what must hold: every live ray is launched, the two LDS schemes agree
    cell by cell, the over-critical core stays exactly zero and energy scales like the 256^3 pass."""
    tr = make_tracer(api, inputs, 512)
    d = tr.derived
    assert (d.nrays, d.nt) == (501264, 2048)
    e3, c3 = run(tr, torch_cuda, kernel_variant=3, stats=True)
    assert c3.rays_traced == 60 * d.nlive_rays
    assert 1.5e10 < c3.ray_steps < 2.0e10
    assert e3[257, 257, 257] == 0.0 and (e3[250:264, 250:264, 250:264] == 0).all()
    total3 = float(e3.sum())
    assert 0.9 < total3 / (1.0076068555e19 * 4.0) < 1.1        # sum(edep) ~ n^2 at fixed rays/zone
    e2, c2 = run(tr, torch_cuda, kernel_variant=2)
    assert c2.ray_steps == c3.ray_steps
    assert parity_err(e3, e2) < 1e-10
    print("512^3: %d ray-steps, %.3f global atomics/step" % (c3.ray_steps, c3.global_atomics / c3.ray_steps))
    tr.close()


@pytest.mark.parametrize("nprofile", [2, 64, 2048])
def test_analytic_profile_of_other_lengths(api, oracle, inputs, torch_cuda, nprofile):
    """Profiles need not have 443 rows (def.cuh:33 is a run-time field here).  SURVEY 8(d)'s synthetic
    stand-in ne(r) = ncrit * exp(-(r - 0.035)/0.01), Te rising with r; 2 rows = a linear profile."""
    bn = inputs[0]
    from cbet_raytracing_3d_amd.tracer import RayTracer
    p = api.default_params(40, nbeams=5, nprofile=nprofile)
    d = api.derive(p)
    r = np.linspace(0.0, 0.3, nprofile)
    if nprofile == 2:       # two rows can only be a straight line: keep it under-critical everywhere
        ne = d.ncrit * np.array([0.9, 0.0])
    else:
        ne = d.ncrit * np.exp(-(r - 0.035) / 0.01)
    te = 500.0 + 6000.0 * r
    beams = [2, 14, 33, 47, 58]
    tr = RayTracer(p, r, ne, te, beam_norm=bn[beams])
    e, c = run(tr, torch_cuda)
    cfg = oracle.default_config(40, nbeams=5, nprofile=nprofile)
    oe, osteps = oracle.trace(cfg, bn[beams].copy(), r, ne, te, nthreads=NCPU)
    assert c.ray_steps == osteps and osteps > 0
    assert parity_err(e, oe) < PARITY_TOL
    tr.close()
    with pytest.raises(api.CbetError):
        api.derive(api.default_params(40, nprofile=2049))


def test_orchestrator_fails_cleanly_without_enough_devices(api, inputs, torch_cuda):
    """cbet_ray_tracing on more devices than exist: a status, not a crash (the reference ignores
    every return value, main.cu:136-151); the next valid call still works."""
    bn, r, ne, te = inputs
    ndev = torch_cuda.cuda.device_count()
    p = api.default_params(24, nbeams=2)
    edep = np.zeros((26, 26, 26))
    with pytest.raises(api.CbetError) as ei:
        api.ray_tracing(te, r, ne, edep, p, beam_norm=bn[:2], gpus=list(range(ndev + 1)))
    assert ei.value.code in (api.ENODEVICE, api.EHIP)
    assert float(np.abs(edep).sum()) == 0.0              # nothing was added on failure
    with pytest.raises(api.CbetError):
        api.ray_tracing(te, r, ne, edep, p, beam_norm=bn[:2], gpus=[-1])
    with pytest.raises(api.CbetError) as ei:             # one rank per device: RCCL cannot pair a device with itself
        api.ray_tracing(te, r, ne, edep, p, beam_norm=bn[:2], gpus=[0, 0])
    assert ei.value.code == api.EINVAL and float(np.abs(edep).sum()) == 0.0
    timers, cnt = api.ray_tracing(te, r, ne, edep, p, beam_norm=bn[:2], gpus=[0])
    assert cnt.ray_steps > 0 and edep.sum() > 0
    # the CLI takes the same path: --gpus beyond the box is an error message and a non-zero exit, not a crash
    import subprocess
    from cbet_raytracing_3d_amd import build
    from conftest import ROOT
    out = subprocess.run([build.CLI_PATH, "10", "--n", "24", "--beams", "2", "--gpus", str(ndev + 1)], cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and ("hipSetDevice" in out.stderr or "device" in out.stderr.lower())


def test_native_wide_index_grid_832(api, inputs, torch_cuda):
    """n = 832: each node table is 4.6 GB, so byte offsets exceed 32 bits and the kernels switch to
    64-bit indexing on their own (the maximum the 32-bit node tags allow is n = 1288).  No oracle at
    this size in test time: the two LDS schemes must agree cell by cell and count the same steps."""
    tr = make_tracer(api, inputs, 832, beams=[0, 21, 40, 59])
    d = tr.derived
    assert 8 * tr.params.nx * tr.params.ny * tr.params.nz >= 2 ** 32
    e3, c3 = run(tr, torch_cuda, kernel_variant=3, stats=True)
    assert c3.rays_traced == 4 * d.nlive_rays and c3.ray_steps > 3e9
    s3, m3, z3 = float(e3.sum()), float(e3.max()), e3[417, 417, 417]
    del e3
    e2, c2 = run(tr, torch_cuda, kernel_variant=2)
    assert c2.ray_steps == c3.ray_steps
    assert abs(float(e2.sum()) / s3 - 1) < 1e-12 and abs(float(e2.max()) / m3 - 1) < 1e-12 and z3 == 0.0
    tr.close()


def test_full_grid_parity_128(api, oracle, inputs, torch_cuda):
    """60 beams, s83177, 128^3: all 130^3 cells against the oracle (the largest size at which the
    oracle runs in test time), default kernel; also pins sharded (1/8) launches cell by cell."""
    bn, r, ne, te = inputs
    tr = make_tracer(api, inputs, 128)
    e, c = run(tr, torch_cuda)
    oe, osteps = oracle.trace(oracle.default_config(128), bn, r, ne, te, nthreads=NCPU)
    assert c.ray_steps == osteps
    err = parity_err(e, oe)
    print("128^3: %d ray-steps, parity err %.3e" % (osteps, err))
    assert err < PARITY_TOL
    assert np.array_equal(e == 0, oe == 0)
    parts = tr.new_grid()
    for s in range(8):
        tr.launch(parts, shard_index=s, shard_count=8)
    assert parity_err(parts.cpu().numpy(), oe) < PARITY_TOL
    tr.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_beam_resolved_deposition(api, oracle, inputs, torch_cuda, variant):
    """per_beam_grids: every beam accumulates into its own grid (what a cross-beam stage consumes).
    Beams are independent in the reference physics, so grid b must equal the oracle's trace of beam b
    alone, and the grids must add up to the ordinary single-grid result."""
    bn, r, ne, te = inputs
    beams = [4, 17, 30, 43, 56]
    tr = make_tracer(api, inputs, 48, beams=beams)
    one, c1 = run(tr, torch_cuda, kernel_variant=variant)
    per = tr.new_grid(per_beam=True)
    tr.counters(reset=True)
    tr.launch(per, kernel_variant=variant)
    c = tr.counters(reset=True)
    per = per.cpu().numpy()
    assert per.shape == (5, 50, 50, 50) and c.ray_steps == c1.ray_steps
    assert parity_err(per.sum(0), one) < 1e-11
    cfg = oracle.default_config(48, nbeams=5)
    for b in range(5):
        oe, _ = oracle.trace(cfg, bn[beams].copy(), r, ne, te, beam_lo=b, beam_hi=b + 1, nthreads=NCPU)
        assert parity_err(per[b], oe) < PARITY_TOL, b
    # sharded + beam range: only the requested beams' grids are touched; the two contiguous halves of the (beam 1,
    # beam 2) list are beam 1 and beam 2
    part = tr.new_grid(per_beam=True)
    tr.launch(part, kernel_variant=variant, beam_lo=1, beam_hi=3, shard_index=1, shard_count=2)
    part = part.cpu().numpy()
    assert float(np.abs(part[[0, 1, 3, 4]]).sum()) == 0.0 and parity_err(part[2], per[2]) < 1e-11
    part = tr.new_grid(per_beam=True)
    tr.launch(part, kernel_variant=variant, beam_lo=1, beam_hi=3, shard_index=0, shard_count=3)   # 2/3 of beam 1
    part = part.cpu().numpy()
    assert float(np.abs(part[[0, 2, 3, 4]]).sum()) == 0.0 and 0 < part[1].sum() < per[1].sum()
    tr.close()


@pytest.mark.parametrize("variant", VARIANTS)
def test_non_spherical_plasma_against_table_oracle(api, oracle, inputs, torch_cuda, variant):
    """SURVEY 8(f) f3 with a genuinely 3-D plasma: an l=2-style density perturbation and an off-centre
    absorbing blob.  The oracle's node-table tracer is the checker (same ray loop as the pinned
    radial path, node values looked up instead of interpolated)."""
    bn, r, ne, te = inputs
    n, beams = 48, [1, 16, 29, 38, 47, 55]
    cfg = oracle.default_config(n, nbeams=len(beams))
    ne3d, kap = oracle.node_tables(cfg, r, ne, te)
    ax = np.linspace(-0.13, 0.13, n)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    R2 = X * X + Y * Y + Z * Z + 1e-30
    ne3d = ne3d * (1.0 + 0.15 * (3.0 * Z * Z / R2 - 1.0) * 0.5 + 0.05 * X * Y / R2)
    kap = kap * (1.0 + 2.0 * np.exp(-((X - 0.05) ** 2 + (Y + 0.04) ** 2 + Z ** 2) / 0.02 ** 2))
    oe, osteps = oracle.trace_tables(cfg, bn[beams].copy(), ne3d, kap, nthreads=NCPU)
    tr = make_tracer(api, inputs, n, beams=beams)
    d = tr.derived
    d_ne, d_kap = torch_cuda.from_numpy(ne3d).cuda(), torch_cuda.from_numpy(kap).cuda()
    e = tr.new_grid()
    tr.counters(reset=True)
    api.trace_nodes(0, d.nindices, d_ne, d_kap, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                    d.xconst, d.yconst, d.zconst,
                    tr.params.copy(beam_lo=0, beam_hi=len(beams), kernel_variant=variant), tr.ctx,
                    torch_cuda.cuda.current_stream().cuda_stream)
    c = tr.counters(reset=True)
    assert c.ray_steps == osteps
    assert parity_err(e.cpu().numpy(), oe) < PARITY_TOL
    sph, _ = run(tr, torch_cuda, kernel_variant=variant)       # the spherical plasma differs visibly
    assert parity_err(sph, oe) > 1e-3
    tr.close()


def test_pass_is_hip_graph_capturable(api, oracle, inputs, torch_cuda):
    """A launch allocates nothing and never synchronises once its workspace exists, so zero +
    tabulate + trace can be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed; the replay
    deposits what the eager launch deposits and what the oracle does."""
    bn, r, ne, te = inputs
    n, beams = 32, list(range(0, 60, 12))
    tr = make_tracer(api, inputs, n, beams=beams)
    e = tr.new_grid()

    def one_pass():
        e.zero_()
        tr.launch(e)

    one_pass()
    torch_cuda.cuda.synchronize()
    eager = e.clone()
    side = torch_cuda.cuda.Stream()
    side.wait_stream(torch_cuda.cuda.current_stream())
    with torch_cuda.cuda.stream(side):
        one_pass()
        graph = torch_cuda.cuda.CUDAGraph()
        with torch_cuda.cuda.graph(graph, stream=side):
            one_pass()
    torch_cuda.cuda.current_stream().wait_stream(side)
    torch_cuda.cuda.synchronize()
    for _ in range(3):
        e.fill_(-1.0)
        graph.replay()
    torch_cuda.cuda.synchronize()
    cfg = oracle.default_config(n, nbeams=len(beams))
    oe, _ = oracle.trace(cfg, bn[beams].copy(), r, ne, te, nthreads=NCPU)
    assert parity_err(e.cpu().numpy(), eager.cpu().numpy()) < PARITY_TOL
    assert parity_err(e.cpu().numpy(), oe) < PARITY_TOL
    del graph
    tr.close()


def test_edep_average_on_device_equals_host(api, inputs, torch_cuda):
    """main.cu:334-349 as a kernel: the 27-point node average, bit for bit the host routine's (same
    summation order), on a ragged grid and on a real deposition grid; and its HBM rate at 256^3."""
    rng = np.random.default_rng(5)
    for (nx, ny, nz) in ((5, 7, 70), (33, 18, 129)):
        e = rng.standard_normal((nx + 2, ny + 2, nz + 2)) * 10.0 ** rng.integers(-3, 12, size=(nx + 2, ny + 2, nz + 2))
        d_e = torch_cuda.from_numpy(e).cuda()
        d_o = torch_cuda.full((nx, ny, nz), float("nan"), dtype=torch_cuda.float64, device="cuda")
        api.edep_average_device(d_e, d_o, nx, ny, nz, torch_cuda.cuda.current_stream().cuda_stream)
        assert np.array_equal(d_o.cpu().numpy(), api.edep_average(e))
    n = 256
    tr = make_tracer(api, inputs, n)
    e = tr.new_grid()
    tr.launch(e)
    out = torch_cuda.empty((n, n, n), dtype=torch_cuda.float64, device="cuda")
    stream = torch_cuda.cuda.current_stream().cuda_stream
    api.edep_average_device(e, out, n, n, n, stream)
    a, b = torch_cuda.cuda.Event(enable_timing=True), torch_cuda.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        api.edep_average_device(e, out, n, n, n, stream)
    b.record()
    torch_cuda.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    gbs = (e.numel() + out.numel()) * 8 / (ms * 1e-3) / 1e9
    print("k_edep_average 256^3: %.3f ms, %.0f GB/s of algorithmic traffic" % (ms, gbs))
    host = api.edep_average(e.cpu().numpy())
    assert np.array_equal(out.cpu().numpy(), host)
    assert gbs > 1000.0
    tr.close()
    with pytest.raises(api.CbetError):
        api.edep_average_device(e, out, 0, n, n, stream)


def test_cli_binary_output_and_cbet_mode(tmp_path):
    """cbet-gpu --npy: what the reference's (dead) save2Hdf5 would have stored -- Coordinate_x/y/z and Edepavg
    [n][n][n] (main.cu:37-94, 321-351) -- plus the haloed edep, as .npy files numpy reads back; and
    --cbet, the CBET iteration behind the same command line."""
    import subprocess
    from cbet_raytracing_3d_amd import api, build
    from conftest import ROOT
    exe, n = build.CLI_PATH, 40
    prefix = str(tmp_path / "run")
    out = subprocess.run([exe, "1", "--n", str(n), "--beams", "6", "--npy", prefix], cwd=ROOT, capture_output=True,
                         timeout=300, text=True)
    assert out.returncode == 0, out.stderr
    edep = np.load(prefix + "_edep.npy")
    avg = np.load(prefix + "_Edepavg.npy")
    assert edep.shape == (n + 2,) * 3 and avg.shape == (n,) * 3 and edep.sum() > 0
    assert np.array_equal(avg, api.edep_average(edep))
    x, y, z = api.node_coordinates(api.default_params(n))
    for name, want in (("x", x), ("y", y), ("z", z)):
        assert np.array_equal(np.load(prefix + "_Coordinate_%s.npy" % name), want)
    steps_plain = int(out.stdout.split("ray-steps ")[1].split()[0])
    out = subprocess.run([exe, "1", "--n", str(n), "--beams", "6", "--cbet", "--npy", prefix + "_cbet"], cwd=ROOT,
                         capture_output=True, timeout=300, text=True)
    assert out.returncode == 0, out.stderr
    assert "converged 1" in out.stdout
    e2 = np.load(prefix + "_cbet_edep.npy")
    ratio = e2.sum() / edep.sum()
    assert 0.8 < ratio < 1.2 and abs(ratio - 1) > 1e-4       # the exchange changes what is absorbed, moderately
    final = int(out.stdout.split("final pass ")[1].split(")")[0])
    assert abs(final - steps_plain) < 0.05 * steps_plain


def test_reference_shaped_driver_through_the_cpp_overloads():
    """tools/cbet_reference_shaped.cpp: rayTracing()'s own call sequence (8 safeGPUAlloc, 7 moveToAndFromGPU,
    launch_ray_XYZ with the reference's thirteen arguments, download, host sum; main.cu:131-210) through the C++
    overloads of include/cbet_reference_api.hpp must give the golden text of `make test` (Makefile:14-17)."""
    import subprocess
    from cbet_raytracing_3d_amd import build
    from conftest import ROOT
    exe = build.REF_SHAPED_PATH
    assert os.path.exists(exe), "cbet-ref-shaped not built"
    out = subprocess.run([exe, "--n", "100", "--print"], cwd=ROOT, capture_output=True, timeout=300)
    assert out.returncode == 0, out.stderr.decode()
    assert len(out.stdout) == 12544620
    assert hashlib.md5(out.stdout).hexdigest() == "cc0909ed1c5938704c51165dc20cb829"
    bad = subprocess.run([exe, "--n", "64", "--gpus", "2"], cwd=ROOT, capture_output=True, timeout=300, text=True)
    assert bad.returncode != 0 and "hipSetDevice(1)" in bad.stdout       # one-GPU box: the helper prints the reason to cout (multi_gpu.cpp:9-27), the driver stops
