"""SweepPipeline (the stream pipeline bench.py times) on one rank: every pass's combined slab, not only the last
one, must equal a plainly traced grid.  With two buffer sets, pass k+2 clears the grid pass k's combine reads: the
clear has to wait for that combine (on one rank a copy on the trace stream)."""
import pytest

from conftest import parity_err

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("overlap,pad_rows", [(False, 1), (True, 1), (True, 0), (False, 120)])
def test_every_pass_of_the_pipeline_is_complete(api, inputs, torch_cuda, overlap, pad_rows):
    """(pad_rows: the pipeline's private grids with rows padded to whole 64-byte lines -- the default --, dense, or an explicit
    pitch: cbet_params.edep_zpitch; the slabs it hands out have the reference's shape either way.)"""
    from cbet_raytracing_3d_amd.tracer import RayTracer, SweepPipeline, traced_pass
    bn, r, ne, te = inputs
    n, nbeams = 96, 12
    tr = RayTracer(api.default_params(n, nbeams=nbeams), r, ne, te, beam_norm=bn[:nbeams])
    want = tr.new_grid()
    traced_pass(tr, want)
    torch_cuda.cuda.synchronize()
    want = want.cpu().numpy()
    pipe = SweepPipeline(tr, 0, 1, overlap_traces=overlap, pad_rows=pad_rows)
    assert pipe.grids[0].shape[-1] == {0: n + 2, 1: 104, 120: 120}[pad_rows] and pipe.slabs[0].shape[-1] == n + 2
    copies = []
    for _ in range(6):
        b = pipe.run_pass()
        pipe.wait_combined(b)                               # this stream waits for the pass's combine, not the host
        copies.append(pipe.slabs[b].clone())
        pipe.release(b)                                     # ... and the combine two passes on waits for this copy
    last = pipe.finish()
    copies.append(last.clone())
    for k, c in enumerate(copies):
        assert parity_err(c.cpu().numpy()[: n + 2], want) < 1e-11, "pass %d" % k
    pipe.close()
    tr.close()


def test_padded_rows_through_the_c_abi(api, inputs, torch_cuda):
    """cbet_params.edep_zpitch at the boundary: a launch into a grid whose rows are 72 doubles long (ragged 40 x 33 x 50 grid,
    all three kernel formulations) deposits exactly what the dense launch deposits and never touches the padding; per-beam
    grids and the CBET hooks refuse a pitch."""
    from cbet_raytracing_3d_amd.tracer import RayTracer
    bn, r, ne, te = inputs
    beams = [3, 17, 31, 44, 58]
    p = api.default_params(40, nbeams=len(beams), rays_per_zone=3)
    p.ny, p.nz = 33, 50
    tr = RayTracer(p, r, ne, te, beam_norm=bn[beams])
    d = tr.derived
    stream = torch_cuda.cuda.current_stream().cuda_stream
    for variant in (1, 2, 3):
        dense = tr.new_grid()
        tr.launch(dense, kernel_variant=variant)
        padded = torch_cuda.full((42, 35, 72), -7.0, dtype=torch_cuda.float64, device="cuda")
        padded[..., :52] = 0.0
        q = tr.params.copy(kernel_variant=variant, edep_zpitch=72, beam_lo=0, beam_hi=len(beams))
        api.launch_ray_XYZ(0, d.nindices, tr.d_te, tr.d_r, tr.d_ne, padded, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                           d.xconst, d.yconst, d.zconst, q, ctx=tr.ctx, stream=stream)
        torch_cuda.cuda.synchronize()
        assert bool((padded[..., 52:] == -7.0).all()), variant
        assert parity_err(padded[..., :52].cpu().numpy(), dense.cpu().numpy()) < 1e-11, variant
    with pytest.raises(api.CbetError) as ei:
        q = tr.params.copy(edep_zpitch=72, per_beam_grids=1, beam_lo=0, beam_hi=len(beams))
        api.launch_ray_XYZ(0, d.nindices, tr.d_te, tr.d_r, tr.d_ne, torch_cuda.zeros((len(beams), 42, 35, 72), dtype=torch_cuda.float64, device="cuda"),
                           tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, d.xconst, d.yconst, d.zconst, q, ctx=tr.ctx, stream=stream)
    assert ei.value.code == api.EINVAL and "edep_zpitch" in str(ei.value)
    tr.close()
