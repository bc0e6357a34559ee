"""Precise timing of the 256^3 trace launch alone (tables kept): N launches timed one by one with HIP events in one process;
prints mean / min / max and the counters.  The library is CBET_LIB_PATH's (variants).  usage: time_trace.py [reps=12] [n=256]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n), r, ne, te)
e = tr.new_grid()
ts = []
for k in range(reps + 3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e.zero_(); tr.counters(reset=True); a.record(); tr.launch(e, stats=bool(int(os.environ.get('CBET_STATS', '0')))); b.record(); torch.cuda.synchronize()
    if k >= 3:
        ts.append(a.elapsed_time(b))
c = tr.counters(reset=True)
print("%s: trace %.3f ms mean, %.3f min, %.3f max over %d launches; edep_sum %.10e steps %d atomics/step %.4f" % (
    os.path.basename(os.environ.get("CBET_LIB_PATH", "shipped")), sum(ts) / len(ts), min(ts), max(ts), len(ts), float(e.sum().item()), c.ray_steps,
    c.global_atomics / max(1, c.ray_steps)))
