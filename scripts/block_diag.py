#!/usr/bin/env python3
"""Diagnostic build of the block kernel (-DCBET_BLOCK_DIAG, CBET_LIB_PATH): which axis the missing lanes fail on, 256^3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
beams = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(60))
r, ne, te = api.load_s83177()
bn = api.omega60_beam_norm()[beams]
p = api.default_params(n, nbeams=len(beams), kernel_variant=api.KERNEL_LDS_BLOCK)
tr = RayTracer(p, r, ne, te, beam_norm=bn)
e = tr.new_grid(); tr.counters(reset=True)
tr.launch(e); torch.cuda.synchronize()
c = tr.counters(reset=True)
print("n %d beams %s: ray-steps %d, missed %.3f%%; failing on x %.3f%% y %.3f%% z %.3f%% of ray-steps; lane-util %.4f" % (
    n, sys.argv[2] if len(sys.argv) > 2 else "all", c.ray_steps, 100.0 * c.lds_evictions / c.ray_steps, 100.0 * (c.wave_steps_miss) / c.ray_steps,
    100.0 * c.wave_steps_wide / c.ray_steps, 100.0 * c.slabs_retired / c.ray_steps, c.ray_steps / (64.0 * c.wave_steps)))
for b in range(len(beams)):
    print("  beam %d dir %s" % (beams[b], bn[b]))
