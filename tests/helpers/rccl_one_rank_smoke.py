"""RCCL on the one GPU of a test box: a ONE-rank process group with backend "nccl" (= RCCL on ROCm) and every
collective of the multi-GPU paths forced to run on it -- all-reduce, reduce-scatter into the slab (SweepPipeline's
combine, on RCCL's stream, overlapped with the next pass), the chunked send/recv exchanges of the slab-owned CBET loop
(a self-exchange through the staging buffers on the communication stream).  RCCL refuses two ranks on one device, so
this is as far as a one-GPU box can take the RCCL code paths: the calls, their stream ordering and their results,
not the xGMI transport.  Results must equal the collective-free single-rank paths.
usage: python tests/helpers/rccl_one_rank_smoke.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_inputs, parity_err  # noqa: E402
from cbet_raytracing_3d_amd import api  # noqa: E402
from cbet_raytracing_3d_amd.tracer import RayTracer, SweepPipeline, allreduce_grid, reduce_scatter_grid, traced_pass  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 200))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
ok = True


def check(name, good, detail=""):
    global ok
    ok &= bool(good)
    print("%-58s %s %s" % (name, "ok" if good else "FAIL", detail), flush=True)


bn, r, ne, te = load_inputs()
n, nbeams = 64, 12
tr = RayTracer(api.default_params(n, nbeams=nbeams), r, ne, te, beam_norm=bn[:nbeams])
want = tr.new_grid()
traced_pass(tr, want)
torch.cuda.synchronize()
ref = want.cpu().numpy()

g = want.clone()
allreduce_grid(g, force=True)
torch.cuda.synchronize()
check("all-reduce on a one-rank RCCL communicator", torch.equal(g, want))

slab = torch.zeros_like(want)
w = reduce_scatter_grid(want, slab, async_op=True, force=True)
if w is not None:
    w.wait()
torch.cuda.synchronize()
check("reduce-scatter (async, RCCL stream) into the slab", torch.equal(slab, want))

pipe = SweepPipeline(tr, 0, 1, overlap_traces=True, force_collectives=True)
pipe.warm()
copies = []
for _ in range(5):
    b = pipe.run_pass()
    pipe.wait_combined(b)                                    # this stream waits for the collective, not the host
    copies.append(pipe.slabs[b].clone())
    pipe.release(b)
last = pipe.finish()
errs = [parity_err(c.cpu().numpy()[: n + 2], ref) for c in copies + [last]]
check("SweepPipeline, 5 passes, RCCL reduce-scatter per pass", max(errs) < 1e-11, "max rel err %.1e" % max(errs))
pipe.close()

beams = [0, 9, 16, 29, 38, 47, 55]
trc = RayTracer(api.default_params(32, nbeams=len(beams)), r, ne, te, beam_norm=bn[beams])
gp = api.default_gain_params(relax=1.0, tolerance=1e-6, max_passes=12)
e0 = trc.new_grid()
rep0 = trc.cbet_solve(e0, gp)
for name, slabs, sparse, opts in (("all-reduce loop", False, False, {}), ("slab loop (dense exchanges: one message per beam, peer and component, all peers of a beam in one grouped send/recv)", True, False, {}),
                                  ("slab loop (sparse exchanges: pack / send-recv / unpack of z-runs)", True, True, {}),
                                  ("slab loop (update split by plane halves, exchange 2 on a second RCCL communicator and stream)", True, False,
                                   dict(slab_layout="halves", two_channels=True))):
    e = trc.new_grid()
    rep = trc.cbet_solve(e, gp, slabs=slabs, force_collectives=True, sparse=sparse, **opts)
    allreduce_grid(e, force=True)
    torch.cuda.synchronize()
    err = parity_err(e.cpu().numpy(), e0.cpu().numpy())
    extra = ""
    if slabs:
        x = rep.get("exchange", {})
        if opts:
            ok &= bool(x.get("two_channels")) and all(len(pcs) == 2 for pcs in rep["slabs"])
        extra = " %s grouped send/recvs, %s messages, %.1f MB sent, staging %d B" % (x.get("chunks"), x.get("messages"), x.get("bytes_sent", 0) / 1e6, x.get("staging_bytes", 0))
    check("CBET " + name, err < 1e-9 and rep["passes"] == rep0["passes"] and rep["converged"],
          "passes %d, max rel err %.1e%s" % (rep["passes"], err, extra))
dist.destroy_process_group()
print("RCCL SMOKE", "PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
