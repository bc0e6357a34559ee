"""CBET stage on the GPU against the CPU restatement of the same model.

PARITY UNPINNED: the reference has no CBET code (def.cuh:94-114 holds unused constants only), so these
tests compare the HIP implementation with oracle/'s restatement of the model DESIGN.md section 9
defines, and check the properties the model promises: hooks off = the reference path, exact pairwise
antisymmetry of the exchange, energy conservation at the fixed point.
"""
import numpy as np
import pytest

from conftest import NCPU, parity_err

pytestmark = pytest.mark.gpu

TOL = 1e-9
N = 32
BEAMS = [0, 16, 29, 38, 47, 55]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "the gpu tests need a HIP device"
    return torch


@pytest.fixture(scope="module")
def api():
    from cbet_raytracing_3d_amd import api as a
    a.lib()
    return a


@pytest.fixture(scope="module")
def setup(api, oracle, inputs, torch_cuda):
    from cbet_raytracing_3d_amd.tracer import RayTracer
    bn, r, ne, te = inputs
    p = api.default_params(N, nbeams=len(BEAMS))
    tr = RayTracer(p, r, ne, te, beam_norm=bn[BEAMS])
    tr.tabulate()
    cfg = oracle.default_config(N, nbeams=len(BEAMS))
    ne3d, kap = oracle.node_tables(cfg, r, ne, te)
    og = oracle.gain_default()
    ofields = np.stack([oracle.trace_cbet(cfg, og, bn[BEAMS].copy(), ne3d, kap, quantity=q, per_beam=True,
                                          nthreads=NCPU)[0] for q in (1, 2, 3, 4)])
    ogain, _ = oracle.gain_field(cfg, og, ofields, ne3d, relax=1.0, nthreads=NCPU)
    yield dict(tr=tr, cfg=cfg, og=og, bn=bn[BEAMS].copy(), ne3d=ne3d, kap=kap, ofields=ofields, ogain=ogain,
               gp=api.default_gain_params(relax=1.0))
    tr.close()


def test_gain_constants_equal_the_oracles(api, oracle, setup):
    assert api.gain_constants(setup["tr"].params, setup["gp"]) == oracle.gain_constants(setup["cfg"], setup["og"])


def test_hooks_off_is_the_reference_path(api, oracle, setup, torch_cuda):
    tr = setup["tr"]
    ref = tr.new_grid()
    tr.counters(reset=True)
    tr.launch(ref)
    c_ref = tr.counters(reset=True)
    e = tr.new_grid()
    bg = torch_cuda.zeros(len(BEAMS), dtype=torch_cuda.float64, device="cuda")
    tr.launch_cbet(e, setup["gp"], gain=None, beam_gain=bg)
    c = tr.counters(reset=True)
    assert c.ray_steps == c_ref.ray_steps
    assert parity_err(e.cpu().numpy(), ref.cpu().numpy()) < TOL
    assert float(bg.abs().sum()) == 0.0
    # a gain field of zeros changes nothing either (x = 0 -> phi = 1 exactly)
    e2 = tr.new_grid()
    tr.launch_cbet(e2, setup["gp"], gain=tr.new_grid(per_beam=True), beam_gain=bg)
    assert tr.counters(reset=True).ray_steps == c_ref.ray_steps
    assert parity_err(e2.cpu().numpy(), ref.cpu().numpy()) < TOL


def test_field_pass_matches_oracle(api, setup, torch_cuda):
    """One fused trace deposits all four per-beam fields; the oracle deposits them one by one."""
    tr = setup["tr"]
    f = tr.new_fields()
    tr.counters(reset=True)
    tr.launch_cbet(f, setup["gp"], fields=True)
    c = tr.counters(reset=True)
    ref = tr.new_grid()
    tr.launch(ref)
    assert c.ray_steps == tr.counters(reset=True).ray_steps     # one trace, not four
    f = f.cpu().numpy()
    for q in range(4):
        for b in range(len(BEAMS)):
            assert parity_err(f[q, b], setup["ofields"][q, b]) < TOL, (q, b)
    d = tr.derived                     # energy x length / step length = the beam's intensity, W/cm^2
    peak = f[0].max() / (2.99792458e10 * d.dt)
    assert 0.3e14 < peak < 4e14
    # the displacement components sit on own nodes only: a subset of where energy was deposited
    assert np.all((f[1:] != 0).any(axis=0) <= (f[0] != 0))
    # the same fields with a gain field of zeros (x = 0 -> phi = 1 exactly)
    f0 = tr.new_fields()
    tr.launch_cbet(f0, setup["gp"], fields=True, gain=tr.new_grid(per_beam=True))
    assert parity_err(f0.cpu().numpy().reshape(-1), f.reshape(-1)) < TOL
    # the energy field alone (what every pass after the direction-building one deposits): component 0 of the above
    for gain in (None, tr.new_grid(per_beam=True)):
        fe = tr.new_fields()
        tr.counters(reset=True)
        tr.launch_cbet(fe[0], setup["gp"], fields="energy", gain=gain)
        assert tr.counters(reset=True).ray_steps == c.ray_steps
        fe = fe.cpu().numpy()
        assert not fe[1:].any()
        for b in range(len(BEAMS)):
            assert parity_err(fe[0, b], setup["ofields"][0, b]) < TOL, b


@pytest.mark.parametrize("symmetric", [False, True])
def test_gain_field_matches_oracle_and_is_antisymmetric(api, oracle, setup, torch_cuda, symmetric):
    """Both gain kernels: the ordered one (the oracle's sum order) and the one that evaluates every
    unordered pair once with the cell's beams staged in LDS (sums grouped by beam tile, pair function as one quotient)."""
    tr, gp = setup["tr"], setup["gp"]
    fields = torch_cuda.from_numpy(setup["ofields"].copy()).cuda()
    gain = tr.new_grid(per_beam=True)
    scratch = torch_cuda.full_like(gain, float("nan")) if symmetric else None   # contents must not matter
    change = torch_cuda.zeros(2, dtype=torch_cuda.float64, device="cuda")
    tr.gain_field(fields, gain, gp, change, scratch=scratch)
    K, want = gain.cpu().numpy(), setup["ogain"]
    scale = np.abs(want).max()
    assert scale > 1.0                       # a non-trivial gain (1/cm)
    assert np.abs(K - want).max() < TOL * scale
    ch = change.cpu().numpy()
    assert abs(ch[0] / ch[1] - 1.0) < 1e-12  # from zero: every |new - old| is |new|
    if not symmetric:
        assert np.array_equal(K, want)       # same sum order as the oracle: bit for bit
    assert abs(ch[1] / np.abs(want).sum() - 1.0) < 1e-9
    # the normalised fields: intensity and wave vectors; what beams exchange in a cell sums to zero
    nf = fields.cpu().numpy()
    inten = np.where(nf[0] > 0, nf[0], 0.0)
    exch = (inten * K).sum(axis=0)
    assert np.abs(exch).max() <= 1e-11 * np.abs(inten * K).sum(axis=0).max()
    kmag = np.sqrt(nf[1] ** 2 + nf[2] ** 2 + nf[3] ** 2)
    present = setup["ofields"][0] > 0                 # where the beam deposited energy the entry is normalised
    touched = setup["ofields"][0] != 0
    d = tr.derived
    assert np.all(kmag[touched] <= d.omega / 2.99792458e10 * (1 + 1e-12))
    assert np.array_equal(nf[:, ~touched], setup["ofields"][:, ~touched])   # untouched entries are left alone
    assert not nf[0][touched & ~present].any()                              # no intensity where the energy is not positive
    # frozen directions: fresh energy on top of the k entries of that call gives the same K
    fields[0] = torch_cuda.from_numpy(setup["ofields"][0].copy()).cuda()
    gain_f = torch_cuda.zeros_like(gain)
    tr.gain_field(fields, gain_f, gp, None, scratch=scratch, frozen=True)
    assert np.abs(gain_f.cpu().numpy() - K).max() < 1e-12 * scale
    assert np.array_equal(fields[1:].cpu().numpy(), nf[1:])                 # ... and leaves them alone
    # under-relaxation: a second call with relax = 0.25 moves a quarter of the way towards the same K
    gain2 = torch_cuda.zeros_like(gain)
    fields2 = torch_cuda.from_numpy(setup["ofields"].copy()).cuda()
    tr.gain_field(fields2, gain2, api.default_gain_params(relax=0.25), None, scratch=scratch)
    assert np.abs(gain2.cpu().numpy() - 0.25 * K).max() < 1e-12 * scale


@pytest.mark.parametrize("nbeams", [60, 64, 3, 1])
def test_pair_once_kernel_on_crowded_cells_equals_the_ordered_kernel(api, inputs, torch_cuda, nbeams):
    """The pair-once kernel stages a 16-cell z-run's present beams in 20 LDS slots and takes a run crossed by more beams in
    halves (<= 40) or quarters (<= 64): synthetic fields of all 60 beams whose crowding grows along x -- ~9, ~30 and ~54
    beams per cell, entries that are touched but not present (E < 0), untouched ones (E = 0), a ragged last z-run
    (nz + 2 = 18 cells: a full run and a run of two) -- must give the ordered kernel's K, and the same normalised fields, first
    call and frozen.  Also with every mask bit in use (64 beams) and with fewer beams than a tile holds (3, 1)."""
    from cbet_raytracing_3d_amd.tracer import RayTracer
    torch = torch_cuda
    bn, r, ne, te = inputs
    n = 16
    tr = RayTracer(api.default_params(n, nbeams=nbeams), r, ne, te, beam_norm=np.concatenate([bn, bn[:4]])[:nbeams])
    tr.tabulate()
    gen = torch.Generator(device="cuda").manual_seed(20261004)
    shape = (nbeams,) + tr.grid_shape
    u = torch.rand(shape, generator=gen, device="cuda", dtype=torch.float64)
    row = torch.rand(shape[:3] + (1,), generator=gen, device="cuda", dtype=torch.float64)   # a beam crosses a z-row or not
    x = torch.arange(tr.grid_shape[0], device="cuda").view(1, -1, 1, 1)
    density = torch.where(x < 5, 0.15, torch.where(x < 10, 0.45, 0.9)).to(torch.float64)
    present = (row < density) & (u < 0.8)
    touched_only = (row < density + 0.03) & ~present & (u < 0.9)
    raw = torch.zeros((4,) + shape, dtype=torch.float64, device="cuda")
    e = torch.rand(shape, generator=gen, device="cuda", dtype=torch.float64) * 1e3 + 1.0
    raw[0] = torch.where(present, e, torch.where(touched_only, -e, torch.zeros_like(e)))
    raw[1:] = (torch.rand((3,) + shape, generator=gen, device="cuda", dtype=torch.float64) - 0.5) * (raw[0] != 0)
    crowd = torch.stack([present[..., :16].any(-1).sum(0), present[..., 16:].any(-1).sum(0)])   # beams per z-run
    if nbeams >= 60:
        assert int((crowd <= 20).sum()) > 50 and int(((crowd > 20) & (crowd <= 40)).sum()) > 50 and int((crowd > 40).sum()) > 50
    gp = api.default_gain_params(relax=1.0)
    results = {}
    for pair_once in (False, True):
        f = raw.clone()
        k = tr.new_grid(per_beam=True)
        ch = torch.zeros(2, dtype=torch.float64, device="cuda")
        tr.gain_field(f, k, gp, ch, pair_once=pair_once)
        f2 = f.clone()
        f2[0] = raw[0]
        k2 = torch.zeros_like(k)
        tr.gain_field(f2, k2, gp, None, pair_once=pair_once, frozen=True)
        results[pair_once] = (f.cpu().numpy(), k.cpu().numpy(), ch.cpu().numpy(), f2.cpu().numpy(), k2.cpu().numpy())
    (f_o, k_o, ch_o, f2_o, k2_o), (f_p, k_p, ch_p, f2_p, k2_p) = results[False], results[True]
    scale = np.abs(k_o).max()
    assert (scale > 0) == (nbeams > 1)
    scale = max(scale, 1e-300)
    assert np.array_equal(f_p, f_o) and np.array_equal(f2_p, f2_o)          # the normalisation is the same arithmetic
    assert np.abs(k_p - k_o).max() < 1e-12 * scale
    assert np.abs(k2_p - k_o).max() < 1e-12 * scale and np.abs(k2_o - k_o).max() < 1e-12 * scale
    if nbeams > 1:
        assert abs(ch_p[1] / ch_o[1] - 1.0) < 1e-12 and abs(ch_p[0] / ch_o[0] - 1.0) < 1e-12
        # exchange in a cell sums to zero, crowded or not
        inten = np.where(f_p[0] > 0, f_p[0], 0.0)
        assert np.abs((inten * k_p).sum(axis=0)).max() <= 1e-11 * np.abs(inten * k_p).sum(axis=0).max()
    else:
        assert not k_p.any() and not ch_p.any()
    tr.close()


def test_all_sixty_beams_fields_and_gain_match_the_oracle(api, oracle, inputs, torch_cuda):
    """The whole OMEGA-60 set on a small grid: runs of 16 cells crossed by up to ~40 beams (the pair-once kernel's halves
    path on real fields), the fused four-component pass and the gain of both kernels against the CPU model."""
    from cbet_raytracing_3d_amd.tracer import RayTracer
    bn, r, ne, te = inputs
    n = 40
    tr = RayTracer(api.default_params(n, nbeams=60), r, ne, te)
    tr.tabulate()
    cfg = oracle.default_config(n, nbeams=60)
    ne3d, kap = oracle.node_tables(cfg, r, ne, te)
    og = oracle.gain_default()
    ofields = np.stack([oracle.trace_cbet(cfg, og, bn.copy(), ne3d, kap, quantity=q, per_beam=True, nthreads=NCPU)[0]
                        for q in (1, 2, 3, 4)])
    ogain, _ = oracle.gain_field(cfg, og, ofields, ne3d, relax=1.0, nthreads=NCPU)
    gp = api.default_gain_params(relax=1.0)
    fields = tr.new_fields()
    tr.launch_cbet(fields, gp, fields=True)
    got = fields.cpu().numpy()
    for q in range(4):
        assert parity_err(got[q].reshape(-1), ofields[q].reshape(-1)) < TOL, q
    crowd = (ofields[0] > 0).sum(0)
    assert crowd.max() > 20                       # some cells see more beams than the kernel has LDS slots for a full run
    scale = np.abs(ogain).max()
    for pair_once in (False, True):
        f = torch_cuda.from_numpy(ofields.copy()).cuda()
        k = tr.new_grid(per_beam=True)
        tr.gain_field(f, k, gp, None, pair_once=pair_once)
        err = np.abs(k.cpu().numpy() - ogain).max()
        assert err < TOL * scale, (pair_once, err / scale)
        if not pair_once:
            assert err == 0.0                     # the ordered kernel: the oracle's operations in the oracle's order
    tr.close()


def test_gain_pass_matches_oracle(api, oracle, setup, torch_cuda):
    tr, gp = setup["tr"], setup["gp"]
    gain = torch_cuda.from_numpy(setup["ogain"].copy()).cuda()
    e = tr.new_grid()
    bg = torch_cuda.zeros(len(BEAMS), dtype=torch_cuda.float64, device="cuda")
    tr.counters(reset=True)
    tr.launch_cbet(e, gp, gain=gain, beam_gain=bg)
    c = tr.counters(reset=True)
    oe, osteps, obg = oracle.trace_cbet(setup["cfg"], setup["og"], setup["bn"], setup["ne3d"], setup["kap"],
                                        gain=setup["ogain"], nthreads=NCPU)
    assert c.ray_steps == osteps
    assert parity_err(e.cpu().numpy(), oe) < TOL
    assert np.abs(bg.cpu().numpy() - obg).max() < TOL * np.abs(obg).max()
    assert np.abs(obg).max() > 1e12          # energy really moves between beams


def test_solve_converges_conserves_and_matches_oracle(api, oracle, setup, torch_cuda):
    tr, cfg, og = setup["tr"], setup["cfg"], setup["og"]
    gp = api.default_gain_params(tolerance=1e-6, max_passes=12, relax=1.0)   # six beams: plain iteration converges
    # oracle: the same fixed-point iteration -- the direction fields of the gain-free first pass are kept
    # (direction_passes = 1), later passes deposit the energy field only
    assert gp.direction_passes == 1
    K, passes, F = None, 0, None
    for it in range(gp.max_passes):
        new = [oracle.trace_cbet(cfg, og, setup["bn"], setup["ne3d"], setup["kap"], gain=K, quantity=q,
                                 per_beam=True, nthreads=NCPU)[0] for q in ((1, 2, 3, 4) if it == 0 else (1,))]
        if it == 0:
            F = np.stack(new)
        else:
            F[0] = new[0]
        K, ch = oracle.gain_field(cfg, og, F, setup["ne3d"], relax=gp.relax, gain=K, nthreads=NCPU)
        passes = it + 1
        if ch[0] / ch[1] < gp.tolerance:
            break
    oe, osteps, obg = oracle.trace_cbet(cfg, og, setup["bn"], setup["ne3d"], setup["kap"], gain=K, nthreads=NCPU)

    # native loop (C ABI) and the rank-aware Python loop must agree with it and with each other
    e = tr.new_grid()
    rep = api.cbet_solve(tr.d_te, tr.d_r, tr.d_ne, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                         tr.params, gp, ctx=tr.ctx, stream=torch_cuda.cuda.current_stream().cuda_stream)
    assert rep.converged == 1 and rep.passes == passes
    assert rep.ray_steps_final == osteps
    assert parity_err(e.cpu().numpy(), oe) < 1e-7     # the iteration compounds rounding differences of the deposits
    bg = np.array(rep.beam_gain[:len(BEAMS)])
    assert np.abs(bg - obg).max() < 1e-7 * np.abs(obg).max()
    assert rep.imbalance < 1e-4                       # what beams gain and lose cancels at the fixed point
    assert abs(obg.sum()) / np.abs(obg).sum() < 1e-4
    e2 = tr.new_grid()
    rep2 = tr.cbet_solve(e2, gp)
    assert rep2["converged"] and rep2["passes"] == passes
    assert parity_err(e2.cpu().numpy(), e.cpu().numpy()) < 1e-7
    # CBET moves energy but the plasma still absorbs a comparable total
    ref = tr.new_grid()
    tr.launch(ref)
    ratio = float(e.sum() / ref.sum())
    assert 0.9 < ratio < 1.1 and abs(ratio - 1.0) > 1e-4


def test_cbet_argument_errors(api, setup, torch_cuda):
    tr = setup["tr"]
    e = tr.new_grid()
    with pytest.raises(api.CbetError) as ei:
        tr.launch_cbet(e, api.default_gain_params(max_exponent=2.0))
    assert ei.value.code == api.EINVAL
    d = tr.derived
    stream = torch_cuda.cuda.current_stream().cuda_stream
    with pytest.raises(api.CbetError) as ei:   # quantity is CBET_DEPOSIT_ENERGY, _FIELDS or _FIELD_ENERGY
        api.trace_cbet(0, d.nindices, None, None, None, 5, e, None, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r,
                       tr.d_phase_r, d.xconst, d.yconst, d.zconst, tr.params.copy(beam_lo=0, beam_hi=len(BEAMS)),
                       setup["gp"], tr.ctx, stream)
    assert ei.value.code == api.EINVAL
    p = tr.params.copy(kernel_variant=1, beam_lo=0, beam_hi=len(BEAMS))
    with pytest.raises(api.CbetError) as ei:   # hooks exist for the default kernel only
        api.trace_cbet(0, d.nindices, None, None, None, api.DEPOSIT_FIELDS, tr.new_fields(), None, tr.d_bbeam_norm,
                       tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, d.xconst, d.yconst, d.zconst, p, setup["gp"], tr.ctx,
                       stream)
    assert ei.value.code == api.EINVAL
    with pytest.raises(api.CbetError) as ei:
        api.cbet_solve(tr.d_te, tr.d_r, tr.d_ne, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                       tr.params.copy(shard_index=0, shard_count=2), setup["gp"], ctx=tr.ctx)
    assert ei.value.code == api.EINVAL
    with pytest.raises(api.CbetError) as ei:   # at least one pass has to build the direction field
        tr.launch_cbet(e, api.default_gain_params(direction_passes=0))
    assert ei.value.code == api.EINVAL


def test_cbet_wide_index_path(api, inputs, setup, torch_cuda):
    """64-bit gathers (grids of 4 GB and more) with the CBET hooks: forced on a small grid, same results."""
    from cbet_raytracing_3d_amd.tracer import RayTracer
    bn, r, ne, te = inputs
    trw = RayTracer(api.default_params(N, nbeams=len(BEAMS), force_wide_index=1), r, ne, te, beam_norm=bn[BEAMS])
    trw.tabulate()
    tr, gp = setup["tr"], setup["gp"]
    gain = torch_cuda.from_numpy(setup["ogain"].copy()).cuda()
    fw, f = trw.new_fields(), tr.new_fields()
    trw.launch_cbet(fw, gp, fields=True, gain=gain)
    tr.launch_cbet(f, gp, fields=True, gain=gain)
    assert parity_err(fw.cpu().numpy().reshape(-1), f.cpu().numpy().reshape(-1)) < TOL
    ew, e = trw.new_grid(), tr.new_grid()
    trw.launch_cbet(ew, gp, gain=gain)
    tr.launch_cbet(e, gp, gain=gain)
    assert parity_err(ew.cpu().numpy(), e.cpu().numpy()) < TOL
    trw.close()


def test_gain_update_by_slabs_equals_the_whole(api, setup, torch_cuda):
    """cbet_gain_field_slab: updating the x-slabs one after the other (ragged split, odd boundaries that cut through
    the kernel's two-plane bricks) gives bit for bit the gain of one update over the whole grid, and the
    convergence sums add up; cells outside a slab are not touched."""
    tr, gp = setup["tr"], setup["gp"]
    whole = tr.new_grid(per_beam=True)
    ch_whole = torch_cuda.zeros(2, dtype=torch_cuda.float64, device="cuda")
    tr.gain_field(torch_cuda.from_numpy(setup["ofields"].copy()).cuda(), whole, gp, ch_whole,
                  scratch=torch_cuda.empty_like(whole))
    for sym in (False, True):
        parts = tr.new_grid(per_beam=True).fill_(-7.0)
        ref = tr.new_grid(per_beam=True).fill_(-7.0)
        tr.gain_field(torch_cuda.from_numpy(setup["ofields"].copy()).cuda(), ref, gp, None,
                      scratch=torch_cuda.empty_like(ref) if sym else None)
        ch = torch_cuda.zeros(2, dtype=torch_cuda.float64, device="cuda")
        fields = torch_cuda.from_numpy(setup["ofields"].copy()).cuda()
        cuts = [0, 1, 12, 13, 29, N + 2]
        for x0, x1 in zip(cuts[:-1], cuts[1:]):
            before = parts.clone()
            tr.gain_field(fields, parts, gp, ch, scratch=torch_cuda.empty_like(parts) if sym else None, x_lo=x0, x_hi=x1)
            assert torch_cuda.equal(parts[:, :x0], before[:, :x0]) and torch_cuda.equal(parts[:, x1:], before[:, x1:])
        assert torch_cuda.equal(parts, ref)
    with pytest.raises(api.CbetError):
        tr.gain_field(fields, parts, gp, None, x_lo=5, x_hi=N + 3)


def test_packed_slab_storage_equals_whole_grid_storage(api, setup, torch_cuda):
    """cbet_gain_field_packed / cbet_params.grid_beam0, grid_beams: arrays that hold only one x-slab of every beam,
    or only some beams' grids, must give what the whole arrays give -- bit for bit for the fields, the ordered gain kernel and
    the traces, to the last bits for the pair-once gain kernel (the storage one rank of the slab-owned loop keeps)."""
    tr, gp = setup["tr"], setup["gp"]
    stream = torch_cuda.cuda.current_stream().cuda_stream
    fields = torch_cuda.from_numpy(setup["ofields"].copy()).cuda()
    x0, x1 = 9, 22
    for sym in (False, True):
        ref = tr.new_grid(per_beam=True)
        f_ref = fields.clone()
        ch_ref = torch_cuda.zeros(2, dtype=torch_cuda.float64, device="cuda")
        tr.gain_field(f_ref, ref, gp, ch_ref, scratch=torch_cuda.empty_like(ref) if sym else None, x_lo=x0, x_hi=x1)
        f_pk = fields[:, :, x0:x1].contiguous()
        g_pk = torch_cuda.zeros_like(ref[:, x0:x1]).contiguous()
        ch = torch_cuda.zeros(2, dtype=torch_cuda.float64, device="cuda")
        api.gain_field_packed(f_pk, None, g_pk, torch_cuda.empty_like(g_pk) if sym else None, ch, x0, x1, tr.params, gp,
                              tr.ctx, stream)
        assert torch_cuda.equal(f_pk, f_ref[:, :, x0:x1])              # the normalisation is cell-local arithmetic
        if sym:
            # the pair-once kernel cuts a z-row into runs that start on 128-byte lines of the ARRAYS, so which cells share a
            # run -- and with it the grouping of a cell's pair sums -- follows the storage: equal to the last bits, like
            # against the ordered kernel
            assert float((g_pk - ref[:, x0:x1]).abs().max()) < 1e-12 * float(ref.abs().max())
        else:
            assert torch_cuda.equal(g_pk, ref[:, x0:x1])
        assert float(((ch - ch_ref).abs() / ch_ref).max()) < 1e-12      # atomically accumulated: order differs
    # beams [2, 5) only: field pass and deposition pass with compact arrays against slices of the full ones
    gain = torch_cuda.from_numpy(setup["ogain"].copy()).cuda()
    full_f, full_e = tr.new_fields(), tr.new_grid()
    tr.launch_cbet(full_f, gp, fields=True, gain=gain, beam_lo=2, beam_hi=5)
    tr.launch_cbet(full_e, gp, gain=gain, beam_lo=2, beam_hi=5)
    part_f = torch_cuda.zeros((4, 3) + tr.grid_shape, dtype=torch_cuda.float64, device="cuda")
    part_e = tr.new_grid()
    gpart = gain[2:5].contiguous()
    tr.launch_cbet(part_f, gp, fields=True, gain=gpart, beam_lo=2, beam_hi=5, grid_beam0=2, grid_beams=3)
    tr.launch_cbet(part_e, gp, gain=gpart, beam_lo=2, beam_hi=5, grid_beam0=2, grid_beams=3)
    assert parity_err(part_f.cpu().numpy().reshape(-1), full_f[:, 2:5].cpu().numpy().reshape(-1)) < TOL
    assert float(full_f[:, :2].abs().sum()) == 0.0 and float(full_f[:, 5:].abs().sum()) == 0.0
    assert parity_err(part_e.cpu().numpy(), full_e.cpu().numpy()) < TOL
    with pytest.raises(api.CbetError):        # beams outside the compact arrays' range
        tr.launch_cbet(part_f, gp, fields=True, beam_lo=1, beam_hi=4, grid_beam0=2, grid_beams=3)


def test_slab_owned_loop_on_one_rank_equals_the_plain_loop(api, setup, torch_cuda):
    """RayTracer.cbet_solve(slabs=True) with world_size 1 (its exchanges are no-ops) must reproduce the plain loop;
    the multi-rank exchanges themselves are covered over gloo in tests/test_cbet_model.py."""
    tr = setup["tr"]
    gp = api.default_gain_params(tolerance=1e-6, max_passes=12, relax=1.0)
    e1, e2 = tr.new_grid(), tr.new_grid()
    r1 = tr.cbet_solve(e1, gp)
    r2 = tr.cbet_solve(e2, gp, slabs=True)
    assert r1["converged"] and r2["converged"] and r1["passes"] == r2["passes"]
    assert parity_err(e2.cpu().numpy(), e1.cpu().numpy()) < 1e-9
    assert np.abs(r1["beam_gain"] - r2["beam_gain"]).max() < 1e-9 * np.abs(r1["beam_gain"]).max()
    assert r2["workspace_bytes"] == api.cbet_slab_workspace_bytes(tr.params, 1, 0) - 8 * (2 + api.MAX_CBET_BEAMS)


def test_slab_loop_as_first_use_of_a_fresh_tracer(api, inputs, torch_cuda):
    """The slab loop's trace groups rotate over four streams that only wait for begin_beams' `ev_ready`: the step-record
    table must be built (not just the node tables) before that event, or -- in a FRESH context, whose record memory is
    uninitialised -- the launches of streams 1..3 read it while k_step_table is still writing it on stream 0.  A solve
    with trace_groups > 1 as the very first use of a new RayTracer must equal the all-reduce loop of another."""
    from cbet_raytracing_3d_amd.tracer import RayTracer
    bn, r, ne, te = inputs
    n, beams = 64, [0, 9, 16, 29, 38, 47, 55, 58]
    gp = api.default_gain_params(tolerance=1e-6, max_passes=6, relax=1.0)
    fresh = RayTracer(api.default_params(n, nbeams=len(beams)), r, ne, te, beam_norm=bn[beams])
    e2 = fresh.new_grid()
    r2 = fresh.cbet_solve(e2, gp, slabs=True, trace_groups=4)      # first thing this context ever runs
    torch_cuda.cuda.synchronize()
    other = RayTracer(api.default_params(n, nbeams=len(beams)), r, ne, te, beam_norm=bn[beams])
    e1 = other.new_grid()
    r1 = other.cbet_solve(e1, gp)
    torch_cuda.cuda.synchronize()
    assert r1["passes"] == r2["passes"]
    assert parity_err(e2.cpu().numpy(), e1.cpu().numpy()) < 1e-9
    fresh.close()
    other.close()


def test_slab_loop_first_pass_fields_survive_the_allocation(api, inputs, torch_cuda):
    """With a communication stream (RCCL, or the stand-in transport used here) the first pass's exchanges run on that
    stream behind their own group's trace only, while the slab arrays are allocated -- zero-filled -- on the caller's
    stream behind ALL traces: the exchanges must also wait for the fill, or the fields that landed early are zeroed.
    One rank, a stand-in transport with nothing to move (the own-part copies still run on the communication stream):
    slab_fields after pass 0 must equal the whole-grid fields of the same beams."""
    from cbet_raytracing_3d_amd.tracer import RayTracer, _DeviceCbetEngine, cbet_fixed_point_slabs
    bn, r, ne, te = inputs
    n, beams = 96, list(range(0, 60, 4))
    tr = RayTracer(api.default_params(n, nbeams=len(beams)), r, ne, te, beam_norm=bn[beams])
    gp = api.default_gain_params(tolerance=1e-30, max_passes=1, relax=1.0)      # exactly one (direction-building) pass
    eng = _DeviceCbetEngine(tr, tr.new_grid(), gp)
    eng.emulate_transport = lambda xch, sends, recvs: None
    snap = {}
    update = eng.update_gain_slab

    def spy(frozen=False):          # (the update normalises the fields in place: look at them just before)
        snap["slab"] = eng.slab_fields[0].clone()
        snap["own"] = eng.own_fields.clone()
        return update(frozen)
    eng.update_gain_slab = spy
    cbet_fixed_point_slabs(eng, gp, len(beams), tr.grid_shape[0], trace_groups=4)
    torch_cuda.cuda.synchronize()
    assert float(snap["own"][0].abs().sum()) > 0.0
    assert torch_cuda.equal(snap["slab"], snap["own"])
    tr.close()


def test_config3_cbet_solve_256_properties(api, inputs, torch_cuda):
    """BASELINE config 3's "full CBET gain iteration": 256^3, 60 beams, the native fixed-point loop with the
    default gain parameters (the run bench.py reports in its `cbet` object).  Parity unpinned (no reference CBET
    code), so the properties: hooks off = the plain pass at this size; the iteration converges inside
    max_passes; what the beams gain and lose cancels (imbalance < 5e-3 at tolerance 1e-4); every pass traces
    every ray; CBET lowers the absorbed total (energy leaves with the outgoing light)."""
    from cbet_raytracing_3d_amd.tracer import RayTracer
    bn, r, ne, te = inputs
    tr = RayTracer(api.default_params(256), r, ne, te, beam_norm=bn)
    gp = api.default_gain_params()
    plain = tr.new_grid()
    tr.counters(reset=True)
    tr.launch(plain)
    c_plain = tr.counters(reset=True)
    assert c_plain.ray_steps == 2123497670
    off = tr.new_grid()
    bg = torch_cuda.zeros(60, dtype=torch_cuda.float64, device="cuda")
    tr.launch_cbet(off, gp, gain=None, beam_gain=bg)          # hooks compiled in, no gain field
    assert tr.counters(reset=True).ray_steps == c_plain.ray_steps
    assert float(bg.abs().sum()) == 0.0
    assert parity_err(off.cpu().numpy(), plain.cpu().numpy()) < TOL
    plain_sum = float(plain.sum())
    del off
    ws = torch_cuda.empty(api.cbet_workspace_bytes(tr.params) // 8, dtype=torch_cuda.float64, device="cuda")
    e = tr.new_grid()
    rep = api.cbet_solve(tr.d_te, tr.d_r, tr.d_ne, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                         tr.params, gp, workspace=ws, ctx=tr.ctx, stream=torch_cuda.cuda.current_stream().cuda_stream)
    torch_cuda.cuda.synchronize()
    assert rep.converged == 1 and 2 <= rep.passes <= gp.max_passes
    assert rep.change < gp.tolerance
    assert rep.imbalance < 5e-3
    assert rep.ray_steps_final > 0 and rep.ray_steps > rep.passes * 1.5e9
    ratio = float(e.sum()) / plain_sum
    assert 0.3 < ratio < 0.95                                   # ~41 % less absorbed with the defaults
    print("256^3 CBET solve: %d passes, change %.2e, imbalance %.2e, absorbed/plain %.3f" %
          (rep.passes, rep.change, rep.imbalance, ratio))
    tr.close()
