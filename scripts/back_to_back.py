"""Is the per-launch constant an artefact of starting each launch on an idle GPU?  Time a 1/8 share
(a) with a host sync before every launch, (b) ten launches enqueued back to back."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
r, ne, te = api.load_s83177()
phases = int(sys.argv[1]) if len(sys.argv) > 1 else -1
tr = RayTracer(api.default_params(256, order_phases=phases), r, ne, te)
print("order_phases", phases)
e = tr.new_grid(); d = tr.derived
stream = torch.cuda.current_stream().cuda_stream
def launch(K, si):
    p = tr.params.copy(beam_lo=0, beam_hi=60, shard_index=si, shard_count=K)
    api.trace_nodes(0, d.nindices, None, None, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                    d.xconst, d.yconst, d.zconst, p, tr.ctx, stream)
api.tabulate_plasma(tr.ctx, tr.params.copy(beam_lo=0, beam_hi=60), tr.d_te, tr.d_r, tr.d_ne, stream)
launch(1, 0); torch.cuda.synchronize()
for K in (8, 4, 1):
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); time.sleep(0.02)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); launch(K, rep % K); t1.record(); torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1))
    a = sum(ts[1:]) / 5
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 16 if K > 1 else 4
    t0.record()
    for rep in range(n): launch(K, rep % K)
    t1.record(); torch.cuda.synchronize()
    b = t0.elapsed_time(t1) / n
    print("1/%d share: %.3f ms after an idle gap, %.3f ms back to back (ideal %.3f)" % (K, a, b, 22.0 / K))
