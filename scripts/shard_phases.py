"""One rank's share (shard_count = K) of the 256^3 sweep for several work-item orders (order_phases)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r, ne, te = api.load_s83177()
for phases in (1, 2, 3, 4, 6, 8):
    tr = RayTracer(api.default_params(n, order_phases=phases), r, ne, te)
    e = tr.new_grid()
    row = []
    for K in (1, 4, 8):
        times = []
        for shard in range(min(K, 2)):
            for rep in range(4):
                e.zero_()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(); tr.launch(e, shard_index=shard, shard_count=K); t1.record(); torch.cuda.synchronize()
                if rep: times.append(t0.elapsed_time(t1))
        row.append(sum(times) / len(times))
    print("order_phases %d: K=1 %.3f ms  K=4 %.3f ms  K=8 %.3f ms" % (phases, *row), flush=True)
    tr.close()
