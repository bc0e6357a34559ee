"""How long does ONE rank's share of the 256^3 sweep take for shard_count = 1,2,4,8 (single GPU)?
Three timings per K:
  * "pass alone": zero the grid + tabulate + step records + trace of the share, nothing beside it (everything but the
    combine) -- a pass that rebuilds the plasma tables, as every pass of the sweep benchmark does;
  * "trace alone": the trace launch of the share on tables that are kept -- a DEPENDENT pass (the CBET iteration: the
    plasma does not change between passes, only the gain does), which cannot overlap with its neighbours either;
  * "pipelined": the steady-state time per pass when consecutive independent passes overlap (tracer.SweepPipeline).
Ideal = t(1)/K; the gap is what caps strong scaling.  usage: shard_timing.py [n=256]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n), r, ne, te)
print("grid %d^3" % n)
e = tr.new_grid()
base = None
for K in (1, 2, 4, 8):
    times = []
    for shard in sorted({0, K // 2, K - 1}):
        for rep in range(4):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(); e.zero_(); tr.launch(e, shard_index=shard, shard_count=K); t1.record(); torch.cuda.synchronize()
            if rep: times.append(t0.elapsed_time(t1))
    t = sum(times) / len(times)
    base = base or t
    print("pass alone, shards %d: %.3f ms per rank and pass (ideal %.3f, efficiency %.1f%%, speed-up before the combine %.2fx)" %
          (K, t, base / K, 100 * base / K / t, base / t))

# the trace launch alone, on kept tables (a dependent pass)
d = tr.derived
stream = torch.cuda.current_stream().cuda_stream
tr.tabulate()
base_t = None
for K in (1, 2, 4, 8):
    times = []
    for shard in sorted({0, K // 2, K - 1}):
        p = tr.params.copy(beam_lo=0, beam_hi=tr.params.nbeams, shard_index=shard, shard_count=K)
        for rep in range(4):
            e.zero_()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            api.trace_nodes(0, d.nindices, None, None, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, d.xconst,
                            d.yconst, d.zconst, p, tr.ctx, stream)
            t1.record(); torch.cuda.synchronize()
            if rep: times.append(t0.elapsed_time(t1))
    t = sum(times) / len(times)
    base_t = base_t or t
    print("trace alone, shards %d: %.3f ms per rank (ideal %.3f, efficiency %.1f%%, speed-up %.2fx)" % (K, t, base_t / K, 100 * base_t / K / t, base_t / t))

# the same shares through the stream pipeline bench.py uses (tracer.SweepPipeline, no process group: the combine is a
# local slab copy), i.e. the steady-state time per pass of one rank when the next pass's tables are prepared beside
# the drain of the current trace
from cbet_raytracing_3d_amd.tracer import SweepPipeline
for K in (1, 2, 4, 8):
    ts = []
    for rank in sorted({0, K // 2, K - 1}):
        pipe = SweepPipeline(tr, rank, K)   # K > 1: consecutive traces overlap (separate streams per buffer set)
        for _ in range(3): pipe.run_pass()
        pipe.finish()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10): pipe.run_pass()
        pipe.finish(); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1) / 10)
        pipe.ctx[1].close()
    t = sum(ts) / len(ts)
    if K == 1: base_p = t
    print("pipelined, shards %d: %.3f ms per rank and pass (speed-up before the combine %.2fx)" % (K, t, base_p / t))
