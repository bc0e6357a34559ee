"""Does the 1/K share suffer from cold node tables?  Time k_trace alone for shard_count = 1, 8 with
(a) a tabulation right before every launch (what a pass does) and (b) tables left warm."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(256), r, ne, te)
e = tr.new_grid(); d = tr.derived
stream = torch.cuda.current_stream().cuda_stream
def once(K, retab, zero):
    p = tr.params.copy(beam_lo=0, beam_hi=60, shard_index=0, shard_count=K)
    if zero: e.zero_()
    if retab: api.tabulate_plasma(tr.ctx, p, tr.d_te, tr.d_r, tr.d_ne, stream)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    api.trace_nodes(0, d.nindices, None, None, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                    d.xconst, d.yconst, d.zconst, p, tr.ctx, stream)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1)
once(1, True, True)
for K in (1, 8):
    for retab, zero in ((True, True), (False, True), (False, False)):
        ts = [once(K, retab, zero) for _ in range(5)][1:]
        print("shards %d  tabulate-before=%-5s zero-before=%-5s : k_trace %.3f ms" % (K, retab, zero, sum(ts) / len(ts)))

def once_beams(lo, hi, K=1, si=0):
    p = tr.params.copy(beam_lo=lo, beam_hi=hi, shard_index=si, shard_count=K)
    e.zero_()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    api.trace_nodes(0, d.nindices, None, None, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                    d.xconst, d.yconst, d.zconst, p, tr.ctx, stream)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1)
full = 22.11
for lo, hi in ((0, 8), (8, 16), (0, 15), (0, 30)):
    ts = [once_beams(lo, hi) for _ in range(4)][1:]
    t = sum(ts) / len(ts)
    print("beams [%d,%d) contiguous: %.3f ms  (ideal %.3f, efficiency %.1f%%)" % (lo, hi, t, full * (hi - lo) / 60, 100 * full * (hi - lo) / 60 / t))
for K in (2, 4, 8):
    ts = [once_beams(0, 60, K, 0) for _ in range(4)][1:]
    t = sum(ts) / len(ts)
    print("interleaved 1/%d: %.3f ms (efficiency %.1f%%)" % (K, t, 100 * full / K / t))
