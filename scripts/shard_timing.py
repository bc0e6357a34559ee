"""How long does ONE rank's share of the 256^3 sweep take for shard_count = 1,2,4,8 (single GPU)?
Ideal = t(1)/K; the gap is tail/launch inefficiency that caps strong scaling."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r, ne, te = api.load_s83177()
order = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tr = RayTracer(api.default_params(n, patch_order=order), r, ne, te)
print("patch_order", order)
e = tr.new_grid()
base = None
for K in (1, 2, 4, 8):
    times = []
    for shard in range(min(K, 3)):
        for rep in range(4):
            e.zero_()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(); tr.launch(e, shard_index=shard, shard_count=K); t1.record(); torch.cuda.synchronize()
            if rep: times.append(t0.elapsed_time(t1))
    t = sum(times) / len(times)
    base = base or t
    # fixed per-step costs on every rank: zero the grid + tabulate (in launch) are inside t already except zero_
    print("shards %d: %.3f ms per rank (ideal %.3f, efficiency %.1f%%)" % (K, t, base / K, 100 * base / K / t))
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(10): e.zero_()
t1.record(); torch.cuda.synchronize()
print("edep.zero_(): %.3f ms" % (t0.elapsed_time(t1) / 10))
