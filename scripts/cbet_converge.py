"""Convergence history of the CBET fixed-point iteration for a few under-relaxation factors or cyclic
schedules of them (a Chebyshev-style cycle 0.49:0.59:0.83 and two-step cycles were tried: no better than a
constant 0.5-0.6, which contracts the change by ~0.57 per pass with 60 beams).
usage: python scripts/cbet_converge.py [n=128] [nbeams=60] [relax,relax,...] [passes=30]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
# each entry is one experiment: a constant factor "0.5" or a cyclic schedule "0.49:0.59:0.83:1.18"
relaxes = [[float(v) for v in x.split(":")] for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [[1.0], [0.5], [0.25]]
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 30
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
tr.tabulate()
fields, gain = tr.new_fields(), tr.new_grid(per_beam=True)
change = torch.zeros(2, dtype=torch.float64, device="cuda")
scratch = torch.empty_like(gain)
bg = torch.zeros(nb, dtype=torch.float64, device="cuda")
e = tr.new_grid()
for sched in relaxes:
    gain.zero_()
    hist = []
    for it in range(passes):
        gp = api.default_gain_params(relax=sched[it % len(sched)])
        fields.zero_()
        tr.launch_cbet(fields, gp, fields=True, gain=gain if it else None)
        change.zero_()
        tr.gain_field(fields, gain, gp, change, scratch=scratch)
        ch = change.cpu().numpy()
        bg.zero_(); e.zero_()
        tr.launch_cbet(e, gp, gain=gain, beam_gain=bg)
        b = bg.cpu().numpy()
        hist.append((ch[0] / ch[1], abs(b.sum()) / np.abs(b).sum(), float(gain.abs().max())))
    print("relax schedule", sched)
    for it, h in enumerate(hist):
        print("  pass %2d change %.3e imbalance %.3e Kmax %.4g" % (it, *h))
