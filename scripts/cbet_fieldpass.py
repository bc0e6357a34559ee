"""Run the fused CBET field pass a few times (driver for rocprofv3 --pmc / --kernel-trace).
usage: python scripts/cbet_fieldpass.py [n=256] [with_gain=1] [reps=2]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with_gain = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, window_stats=1), r, ne, te)
gp = api.default_gain_params()
tr.tabulate()
f = tr.new_fields()
g = tr.new_grid(per_beam=True).fill_(1.0) if with_gain else None
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tr.launch_cbet(f, gp, fields=True, gain=g)
tr.counters(reset=True)
a.record()
for _ in range(reps):
    tr.launch_cbet(f, gp, fields=True, gain=g)
b.record()
torch.cuda.synchronize()
print("field pass: %.2f ms (n=%d, gain=%d)" % (a.elapsed_time(b) / reps, n, with_gain))
c = tr.counters()
print("ray-steps %d lane-atomics/step %.3f wave-steps %d miss-wave-step frac %.4f" %
      (c.ray_steps, c.global_atomics / c.ray_steps, c.wave_steps, c.wave_steps_miss / c.wave_steps))
