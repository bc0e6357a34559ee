"""How long does ONE rank's share of the 256^3 sweep take for shard_count = 1,2,4,8 (single GPU)?
Per rank and pass: zero the grid + tabulate + step records + trace of the share (everything but the combine).
Ideal = t(1)/K; the gap is what caps strong scaling.  usage: shard_timing.py [n=256] [order_phases=-1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
phases = int(sys.argv[2]) if len(sys.argv) > 2 else -1
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, order_phases=phases), r, ne, te)
print("grid %d^3, order_phases %d" % (n, phases))
e = tr.new_grid()
base = None
for K in (1, 2, 4, 8):
    times = []
    for shard in range(min(K, 3)):
        for rep in range(4):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(); e.zero_(); tr.launch(e, shard_index=shard, shard_count=K); t1.record(); torch.cuda.synchronize()
            if rep: times.append(t0.elapsed_time(t1))
    t = sum(times) / len(times)
    base = base or t
    print("shards %d: %.3f ms per rank and pass (ideal %.3f, efficiency %.1f%%, speed-up before the combine %.2fx)" %
          (K, t, base / K, 100 * base / K / t, base / t))
