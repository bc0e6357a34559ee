"""Full-grid parity at the headline size: the 256^3, 60-beam pass on the GPU against the CPU oracle run on the
GPU box's host cores (all (n+2)^3 cells, SURVEY 8(c) metric), plus the same for the CBET iteration's first
field pass.  The test-suite checks full grids up to 128^3 and properties at 256^3; this is the one-off
evidence for the size the metric is quoted on (takes ~1 min of host time)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import parity_err
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
from oracle import cbet_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = min(16, len(os.sched_getaffinity(0)))
r, ne, te = api.load_s83177()
bn = api.omega60_beam_norm()
tr = RayTracer(api.default_params(n), r, ne, te)
e = tr.new_grid(); tr.counters(reset=True)
tr.launch(e); torch.cuda.synchronize()
c = tr.counters(reset=True)
t0 = time.time()
cfg = O.default_config(n)
oe, osteps = np.zeros(O.grid_shape(cfg)), 0
for lo in range(0, 60, 6):          # in chunks of beams, so a long run keeps reporting progress
    _, st = O.trace(cfg, bn.copy(), r, ne, te, beam_lo=lo, beam_hi=lo + 6, nthreads=threads, edep=oe)
    osteps += st
    print("  oracle beams %d-%d done, %.0f s" % (lo, lo + 5, time.time() - t0), flush=True)
print("oracle: %d ray-steps in %.1f s on %d threads" % (osteps, time.time() - t0, threads), flush=True)
g = e.cpu().numpy()
print("GPU: %d ray-steps; equal counts: %s" % (c.ray_steps, c.ray_steps == osteps))
print("max rel err over all %d cells (SURVEY 8(c) metric): %.3e   (BASELINE bound 1e-4)" % (g.size, parity_err(g, oe)))
print("sum %.13e vs %.13e; non-zero cells %d vs %d; exact zeros preserved: %s" %
      (g.sum(), oe.sum(), np.count_nonzero(g), np.count_nonzero(oe), bool(np.array_equal(g == 0, oe == 0))))
