"""Where the first field pass's global atomics come from: the window counters (cbet_params.window_stats) of the four-component
pass (k_trace_window<8,.,4>) beside those of the energy-field pass (<16,.,2>) at the same size.
usage: python scripts/cbet_first_pass_stats.py [n=256] [nbeams=60]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api                      # noqa: E402
from cbet_raytracing_3d_amd.tracer import RayTracer         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
gp = api.default_gain_params()
tr.tabulate()
fields = tr.new_fields()
tr.params.window_stats = 1
for name, kw, out in (("four-component first pass <8,.,4>", dict(fields=True), fields), ("energy-field pass <16,.,2>", dict(fields="energy"), fields[0])):
    out.zero_()
    tr.counters(reset=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    tr.launch_cbet(out, gp, **kw)
    b.record()
    torch.cuda.synchronize()
    c = tr.counters(reset=True)
    ws, rs = float(c.wave_steps), float(c.ray_steps)
    print("%s: %.2f ms (counting build), %d ray-steps, %.4g wave-steps" % (name, a.elapsed_time(b), c.ray_steps, ws))
    print("   lane-level global atomics %.4g = %.3f per ray-step | ray-steps bound for HBM (outside both boxes) %.4g = %.2f %%"
          % (c.global_atomics, c.global_atomics / rs, c.lds_evictions, 100 * c.lds_evictions / rs))
    print("   wave-steps with box B active %.1f %% | with a lane outside both boxes %.1f %% | planes retired %.4g = %.3f per wave-step"
          % (100 * c.wave_steps_wide / ws, 100 * c.wave_steps_miss / ws, c.slabs_retired, c.slabs_retired / ws))
