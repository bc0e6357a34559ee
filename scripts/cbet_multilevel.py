"""Feasibility of a coarse-grid start for the CBET fixed point: converge the gain on a coarse grid first (an eighth or less
of the work per iteration), interpolate it to the fine grid and count the fine iterations that remain -- against the plain
iteration from zero.  The fixed point is the fine grid's either way (same directions: the gain-free first pass; same
tolerance).  usage: python scripts/cbet_multilevel.py [n_fine=256] [n_coarse=128] [nbeams=60]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 60
r, ne, te = api.load_s83177()
gp = api.default_gain_params()           # relax 0.5, tolerance 1e-4
TOL = gp.tolerance


def iterate(tr, gain, max_passes=40, label=""):
    """Directions from a gain-free four-component pass, then energy-field pass + gain update until the change is below TOL,
    starting from `gain` (modified in place).  Returns (passes, seconds, change history)."""
    fields = tr.new_fields()
    change = torch.zeros(2, dtype=torch.float64, device="cuda")
    tr.tabulate()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    start_zero = not bool(gain.any())
    tr.launch_cbet(fields, gp, fields=True)                       # builds k; its energy field is the K = 0 one
    hist = []
    if start_zero:                                                # the plain iteration: the first update uses that energy field
        change.zero_(); tr.gain_field(fields, gain, gp, change, pair_once=True)
        c = change.cpu().numpy(); hist.append(c[0] / c[1])
    else:                                                         # keep `gain`: normalise the directions only (relax -> no change of K)
        keep = gain.clone()
        tr.gain_field(fields, gain, gp, None, pair_once=True)
        gain.copy_(keep); del keep
    while len(hist) < max_passes and (not hist or hist[-1] >= TOL):
        fields[0].zero_()
        tr.launch_cbet(fields[0], gp, fields="energy", gain=gain)
        change.zero_(); tr.gain_field(fields, gain, gp, change, pair_once=True, frozen=True)
        c = change.cpu().numpy(); hist.append(c[0] / c[1])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    del fields
    torch.cuda.empty_cache()
    print("%s: %d gain updates, %.1f ms; change history %s" % (label, len(hist), 1e3 * dt, " ".join("%.1e" % h for h in hist)), flush=True)
    return len(hist), dt


def prolong(gc, n_c, n_f):
    """[nb][(n_c+2)^3] -> [nb][(n_f+2)^3], trilinear in node coordinates (haloed index h <-> node h - 1)."""
    out = gc
    for ax in (1, 2, 3):
        hf = torch.arange(n_f + 2, device="cuda", dtype=torch.float64)
        pos = (hf - 1.0) * (n_c - 1.0) / (n_f - 1.0) + 1.0
        pos = pos.clamp(0.0, n_c + 1.0)
        i0 = pos.floor().clamp(max=n_c).long()
        w = (pos - i0.to(torch.float64))
        shape = [1, 1, 1, 1]; shape[ax] = -1
        a, b = out.index_select(ax, i0), out.index_select(ax, i0 + 1)
        out = a + (b - a) * w.view(shape)
        del a, b
    return out.contiguous()


trf = RayTracer(api.default_params(nf, nbeams=nb), r, ne, te)
g0 = trf.new_grid(per_beam=True)
p_plain, t_plain = iterate(trf, g0, label="plain, %d^3 from zero" % nf)
ref = g0.clone()
del g0
trc = RayTracer(api.default_params(nc, nbeams=nb), r, ne, te)
gc = trc.new_grid(per_beam=True)
p_c, t_c = iterate(trc, gc, label="coarse, %d^3 from zero" % nc)
torch.cuda.synchronize(); t0 = time.perf_counter()
gf = prolong(gc, nc, nf)
torch.cuda.synchronize(); t_p = time.perf_counter() - t0
print("prolongation %.1f ms; interpolated coarse gain against the fine fixed point: sum|dK| / sum|K| = %.3e" % (1e3 * t_p, float((gf - ref).abs().sum() / ref.abs().sum())))
p_f, t_f = iterate(trf, gf, label="fine, %d^3 from the interpolated coarse gain" % nf)
print("fine result against the plain one: sum|dK| / sum|K| = %.3e" % float((gf - ref).abs().sum() / ref.abs().sum()))
print("plain %d updates %.1f ms   |   coarse %d + fine %d updates: %.1f + %.1f + %.1f = %.1f ms" % (p_plain, 1e3 * t_plain, p_c, p_f, 1e3 * t_c, 1e3 * t_p, 1e3 * t_f, 1e3 * (t_c + t_p + t_f)))
