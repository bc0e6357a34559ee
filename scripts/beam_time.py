"""Does a beam's DIRECTION decide how fast it traces?  Every OMEGA beam alone (one launch each, 256^3, padded grid), three
times; printed with its direction and sorted by time."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
r, ne, te = api.load_s83177()
bn = api.omega60_beam_norm()
tr = RayTracer(api.default_params(256, window_stats=1), r, ne, te)   # (the window diagnostics are counted on request)
e = tr.new_grid(zpitch=True)
res = []
for b in range(60):
    ts = []
    for rep in range(4):
        x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e.zero_(); tr.counters(reset=True); x.record(); tr.launch(e, beam_lo=b, beam_hi=b + 1); y.record(); torch.cuda.synchronize()
        if rep: ts.append(x.elapsed_time(y))
    c = tr.counters(reset=True)
    res.append((min(ts), b, c.ray_steps, c.wave_steps, c.slabs_retired / c.wave_steps, 100.0 * c.lds_evictions / c.ray_steps))
for t, b, st, ws, mv, miss in sorted(res):
    print("beam %2d  dir (%+.2f %+.2f %+.2f)  %.3f ms  %.3e ray-steps  %.2f ns per wave-step  moves/wave-step %.3f  misses %.2f %%" % (b, *bn[b], t, st, 1e6 * t / ws, mv, miss))
a = np.array([[r_[0], *np.abs(bn[r_[1]])] for r_ in res])
for k, name in enumerate("xyz"):
    print("correlation of the time with |n_%s|: %+.2f" % (name, np.corrcoef(a[:, 0], a[:, 1 + k])[0, 1]))
