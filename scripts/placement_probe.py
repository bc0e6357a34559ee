"""Does the pass time depend on WHERE the tables and the grid land in memory?  Several contexts (node tables, step records)
and grids are allocated one after the other in ONE process -- the earlier ones kept alive, so each lands elsewhere -- and the
256^3 trace launch is timed on each; then every (context, grid) pair again in a second round."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
r, ne, te = api.load_s83177()
p = api.default_params(256)
trs, grids = [], []


def timed(tr, e, reps=6):
    ts = []
    for k in range(reps + 2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e.zero_(); a.record(); tr.launch(e); b.record(); torch.cuda.synchronize()
        if k >= 2:
            ts.append(a.elapsed_time(b))
    return sum(ts) / len(ts), min(ts)


for i in range(K):
    tr = RayTracer(p, r, ne, te)
    e = tr.new_grid()
    trs.append(tr); grids.append(e)
    a, b = tr.ctx.tables()
    m, lo = timed(tr, e)
    print("context %d: ne3d at 0x%x kappa3d at 0x%x grid at 0x%x : %.3f ms mean, %.3f min" % (i, a, b, e.data_ptr(), m, lo), flush=True)
print("second round, every context with every grid:")
for i, tr in enumerate(trs):
    print("context %d: " % i + "  ".join("%.2f" % timed(tr, e, 3)[0] for e in grids), flush=True)
