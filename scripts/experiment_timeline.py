"""Occupancy timeline of one k_trace launch from per-wave start/end stamps (diagnostic build)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(256), r, ne, te)
e = tr.new_grid(); d = tr.derived
stream = torch.cuda.current_stream().cuda_stream
nwg = (60 * (len(api.live_ray_list(tr.params)) // 64) + K - 1) // K
buf = torch.zeros(3 * nwg + 64, dtype=torch.int64, device="cuda")
os.environ["CBET_TIMELINE_PTR"] = str(buf.data_ptr())
p = tr.params.copy(beam_lo=0, beam_hi=60, shard_index=0, shard_count=K)
api.tabulate_plasma(tr.ctx, p, tr.d_te, tr.d_r, tr.d_ne, stream)
for rep in range(3):
    buf.zero_()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    api.trace_nodes(0, d.nindices, None, None, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                    d.xconst, d.yconst, d.zconst, p, tr.ctx, stream)
    t1.record(); torch.cuda.synchronize()
ms = t0.elapsed_time(t1)
t = buf[:3 * nwg].cpu().numpy().reshape(-1, 3).astype(np.float64)
st, en, steps = t[:, 0], t[:, 1], t[:, 2]
tick = 1e-8 * 1e3      # s_memrealtime: 100 MHz -> ms per tick
t00 = st.min(); st = (st - t00) * tick; en = (en - t00) * tick
print("launch 1/%d: %d waves, event time %.3f ms, stamps span %.3f ms" % (K, nwg, ms, en.max()))
print("wave lifetime ms: mean %.3f  p50 %.3f  p90 %.3f  max %.3f ; steps mean %.0f max %.0f" % (
    (en - st).mean(), np.median(en - st), np.percentile(en - st, 90), (en - st).max(), steps.mean(), steps.max()))
print("last wave dispatched at %.3f ms; waves started in first 0.05 ms: %d" % (st.max(), int((st < 0.05).sum())))
edges = np.linspace(0, en.max(), 25)
for a_, b_ in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a_ + b_)
    active = int(((st <= mid) & (en > mid)).sum())
    print("  t=%.2f ms  active waves %5d %s" % (mid, active, "#" * (active // 100)))
late = np.argsort(en)[-5:]
print("last finishers: ", [(int(i), round(float(st[i]), 3), round(float(en[i]), 3), int(steps[i])) for i in late])
