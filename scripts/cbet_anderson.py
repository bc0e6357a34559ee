"""Prototype: does Anderson(m) mixing of the gain coefficient cut the number of CBET passes?  (torch ops on
the device arrays around the library's field pass and gain kernel; a study script, not the product loop.)
usage: python scripts/cbet_anderson.py [n=128] [nbeams=60] [passes=16]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 16
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
tr.tabulate()
gp1 = api.default_gain_params(relax=1.0)
fields = tr.new_fields()
scratch = tr.new_grid(per_beam=True)

def F(K, first):
    """one pass: fields traced with gain K -> the gain coefficient they imply"""
    fields.zero_()
    tr.launch_cbet(fields, gp1, fields=True, gain=None if first else K)
    out = K.clone()
    tr.gain_field(fields, out, gp1, None, scratch=scratch)   # relax 1: out = raw K
    return out

def run(name, beta, m):
    K = tr.new_grid(per_beam=True)
    hist_K, hist_r = [], []
    print(name)
    for it in range(passes):
        FK = F(K, it == 0)
        rk = FK - K
        change = float(rk.abs().sum() / FK.abs().sum())
        print("  pass %2d  |F(K)-K|/|F(K)| = %.3e" % (it, change), flush=True)
        if change < 1e-4:
            break
        if it == 0:
            Knew = K + rk            # from zero: take the full step
        elif m == 0 or not hist_r:
            Knew = K + beta * rk
        else:
            # Anderson(m): least squares over the last m residual differences
            dR = [rk - rp for rp in hist_r[-m:]]
            dK = [K - kp for kp in hist_K[-m:]]
            A = torch.tensor([[float((a * b).sum()) for b in dR] for a in dR], dtype=torch.float64)
            bvec = torch.tensor([float((a * rk).sum()) for a in dR], dtype=torch.float64)
            gam = torch.linalg.solve(A + 1e-12 * torch.eye(len(dR), dtype=torch.float64) * A.diagonal().max(), bvec)
            Knew = K + beta * rk
            for g, dk, dr in zip(gam.tolist(), dK, dR):
                Knew -= g * (dk + beta * dr)
        hist_K.append(K); hist_r.append(rk)
        hist_K, hist_r = hist_K[-max(m, 1):], hist_r[-max(m, 1):]
        K = Knew
    return it + 1

res = {}
res["damped 0.5"] = run("plain damped iteration, beta 0.5", 0.5, 0)
res["anderson(1) beta 0.5"] = run("Anderson(1), beta 0.5", 0.5, 1)
res["anderson(1) beta 1.0"] = run("Anderson(1), beta 1.0", 1.0, 1)
res["anderson(2) beta 0.7"] = run("Anderson(2), beta 0.7", 0.7, 2)
print(res)
