"""Time the gain kernel (first call and frozen-direction calls) with the library in CBET_LIB_PATH and print checksums of
the gain it produces, so that kernel variants (build.py --variant) can be compared.  usage: gain_variants.py [n=256] [nbeams=60] [hist]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
gp = api.default_gain_params()
fields, gain = tr.new_fields(), tr.new_grid(per_beam=True)
scratch = torch.empty_like(gain)
change = torch.zeros(2, dtype=torch.float64, device="cuda")
tr.tabulate()
tr.launch_cbet(fields, gp, fields=True)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    X, Y, Z = tr.grid_shape
    pres = (fields[0] > 0)
    cells = pres.sum(0).flatten().float()
    print("beams present per cell: mean %.1f, mean over cells with any %.1f, max %d; sum n^2 weighted mean %.1f"
          % (cells.mean(), cells[cells > 0].mean(), int(cells.max()), float((cells ** 3).sum() / (cells ** 2).sum())))
    for (sx, sy, sz) in ((1, 1, 16), (2, 4, 8)):
        px, py, pz = (-X) % sx, (-Y) % sy, (-Z) % sz
        p = torch.nn.functional.pad(pres, (0, pz, 0, py, 0, px))
        b = p.view(nb, (X + px) // sx, sx, (Y + py) // sy, sy, (Z + pz) // sz, sz).amax((2, 4, 6)).sum(0).flatten().float()
        h = torch.histc(b, bins=13, min=0, max=65)
        print("brick %dx%dx%d: beams present per brick mean %.1f (non-empty %.1f), max %d, pair-weighted mean %.1f; histogram by 5: %s"
              % (sx, sy, sz, b.mean(), b[b > 0].mean(), int(b.max()), float((b ** 3).sum() / (b ** 2).sum()), [int(v) for v in h]))
    del pres
raw = fields.clone()


def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fields.copy_(raw); torch.cuda.synchronize()
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), sum(ts) / len(ts)


gain.zero_()
t = timed(lambda: tr.gain_field(fields, gain, gp, change, scratch=scratch), 1)
print("first call (normalise + gain): %.2f ms" % t[0])
normal = fields.clone()
k1 = gain.clone()
energy = raw[0]


def frozen():
    tr.gain_field(fields, gain, gp, change, scratch=scratch, frozen=True)


ts = []
for _ in range(4):
    fields.copy_(normal); fields[0].copy_(energy); gain.copy_(k1); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); frozen(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print("frozen call: min %.2f ms, mean of last three %.2f ms" % (min(ts), sum(ts[1:]) / 3))
print("checksums: sum %.15e  abs-sum %.15e  max %.15e" % (float(gain.sum()), float(gain.abs().sum()), float(gain.abs().max())))
