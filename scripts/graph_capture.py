"""Capture zero + tabulate + trace of one pass in a HIP graph (torch.cuda.CUDAGraph) and replay it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n), r, ne, te)
e = tr.new_grid()
def one_pass():
    e.zero_(); tr.launch(e)
for _ in range(3): one_pass()
torch.cuda.synchronize(); ref = e.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    one_pass()                      # warm-up on the side stream
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        one_pass()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
e.fill_(-1.0); g.replay(); torch.cuda.synchronize()
err = float(((e - ref).abs() / ref.abs().clamp_min(1e-9 * float(ref.abs().max()))).max())
print("graph replay vs eager: max rel err %.2e" % err)
for name, fn in (("eager", one_pass), ("graph", g.replay)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize(); print("%s: %.3f ms per pass at %d^3" % (name, (time.perf_counter() - t0) * 5, n))
