"""k_trace time vs launch size (number of beams traced in one launch): slope = throughput,
intercept = per-launch latency floor (one bundle's lifetime at low load)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(256), r, ne, te)
e = tr.new_grid(); d = tr.derived
stream = torch.cuda.current_stream().cuda_stream
p0 = tr.params.copy(beam_lo=0, beam_hi=60)
api.tabulate_plasma(tr.ctx, p0, tr.d_te, tr.d_r, tr.d_ne, stream)
def t_launch(lo, hi, K=1):
    p = tr.params.copy(beam_lo=lo, beam_hi=hi, shard_index=0, shard_count=K)
    ts = []
    for _ in range(4):
        e.zero_()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        api.trace_nodes(0, d.nindices, None, None, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                        d.xconst, d.yconst, d.zconst, p, tr.ctx, stream)
        t1.record(); torch.cuda.synchronize(); ts.append(t0.elapsed_time(t1))
    return sum(ts[1:]) / 3
bpb = (len(api.live_ray_list(tr.params)) // 64)
print("bundles per beam", bpb, " resident capacity 4096 waves")
for K, label in ((1620, "1 bundle"), (202, "8 bundles"), (25, "65 bundles"), (6, "270 bundles"), (2, "810 bundles")):
    print("%-12s (beam 0, 1/%d share): %.3f ms" % (label, K, t_launch(0, 1, K)))
for nb in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 60):
    t = t_launch(0, nb)
    print("%2d beams (%6d bundles): %.3f ms   %.4f ms/beam" % (nb, nb * bpb, t, t / nb))
