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


@pytest.mark.parametrize("overlap", [False, True])
def test_every_pass_of_the_pipeline_is_complete(api, inputs, torch_cuda, overlap):
    from cbet_raytracing_3d_amd.tracer import RayTracer, SweepPipeline, traced_pass
    bn, r, ne, te = inputs
    n, nbeams = 96, 12
    tr = RayTracer(api.default_params(n, nbeams=nbeams), r, ne, te, beam_norm=bn[:nbeams])
    want = tr.new_grid()
    traced_pass(tr, want)
    torch_cuda.cuda.synchronize()
    want = want.cpu().numpy()
    pipe = SweepPipeline(tr, 0, 1, overlap_traces=overlap)
    copies = []
    for _ in range(6):
        b = pipe.run_pass()
        with torch_cuda.cuda.stream(pipe.s_trace[b]):      # stream-ordered behind this pass's combine
            copies.append(pipe.slabs[b].clone())
    last = pipe.finish()
    copies.append(last.clone())
    for k, c in enumerate(copies):
        assert parity_err(c.cpu().numpy()[: n + 2], want) < 1e-11, "pass %d" % k
    pipe.close()
    tr.close()
