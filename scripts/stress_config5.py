"""BASELINE config 5: 512^3, 60 beams x ~1e6 rays/beam (rays_per_zone = 6), one pass, one GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
r, ne, te = api.load_s83177()
p = api.default_params(512, rays_per_zone=6)
d = api.derive(p)
print("nrays/beam %d (live %d), nt %d, edep %.2f GB, tables %.2f GB" % (d.nrays, d.nlive_rays, d.nt, 8 * d.edep_size / 1e9, 16 * 512 ** 3 / 1e9))
tr = RayTracer(p, r, ne, te)
e = tr.new_grid()
for rep in range(3):
    e.zero_(); tr.counters(reset=True)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(); tr.launch(e); t1.record(); torch.cuda.synchronize()
    c = tr.counters(reset=True)
ms = t0.elapsed_time(t1)
print("pass: %.1f ms, %d ray-steps, %.3e ray-steps/s, %.3f global atomics/step, lane util %.3f, window miss %.4f" % (
    ms, c.ray_steps, c.ray_steps / ms * 1e3, c.global_atomics / c.ray_steps, c.ray_steps / (64.0 * c.wave_steps), c.lds_evictions / c.ray_steps))
print("sum(edep) %.6e  max %.6e" % (float(e.sum()), float(e.max())))
