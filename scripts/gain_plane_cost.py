import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd import tracer as T
n, nb = 256, 60
r, ne, te = api.load_s83177()
tr = T.RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
gp = api.default_gain_params()
X, Y, Z = tr.grid_shape
edep = tr.new_grid()
ws = torch.empty(api.cbet_workspace_bytes(tr.params) // 8, dtype=torch.float64, device="cuda")
api.cbet_solve(tr.d_te, tr.d_r, tr.d_ne, edep, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, tr.params, gp, workspace=ws, ctx=tr.ctx, stream=torch.cuda.current_stream().cuda_stream)
hs = X * Y * Z
fields = ws[: 4 * nb * hs].view((4, nb) + tr.grid_shape)
gain = ws[4 * nb * hs: 5 * nb * hs].view((nb,) + tr.grid_shape)
fields[0].zero_(); tr.launch_cbet(fields[0], gp, fields="energy", gain=gain)
energy = fields[0].clone()
cnt = (energy != 0).sum(0).double()
change = torch.zeros(2, dtype=torch.float64, device="cuda")
def timed(fn, reps=3):
    fn(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / reps
"""What a plane of the CBET gain update costs, against the beams present in it: the pair-once kernel (k_gain_field_sym) timed on
8-plane slabs across the 256^3 / 60-beam grid with converged fields, beside moments of n = beams per node and of m = the most
beams in any cell of a 16-cell z-run (the kernel's unit of work: its cells run in lockstep).  The fit at the end is what
tracer.gain_update_weights uses to cut the slabs of the slab-owned loop."""
zp = (-Z) % 16
m = torch.nn.functional.pad(cnt, (0, zp)).view(X, Y, (Z + zp) // 16, 16).max(-1).values      # per z-run
rows = []
print("x0 planes  ms/plane   mean n   mean n^2   frac n>20   mean m   mean m^2   mean m^3")
for x0 in range(1, 129, 8):
    x1 = x0 + 8
    def f():
        fields[0, :, x0:x1].copy_(energy[:, x0:x1])
        tr.gain_field(fields, gain, gp, change, pair_once=True, frozen=True, x_lo=x0, x_hi=x1)
    tc = timed(lambda: fields[0, :, x0:x1].copy_(energy[:, x0:x1]))
    t = timed(f) - tc
    c = cnt[x0:x1]
    mm = m[x0:x1]
    rows.append((t / 8, 1.0, mm.mean().item(), (mm ** 2).mean().item(), (mm ** 3).mean().item()))
    print("%3d %3d   %.4f   %.2f   %.1f   %.3f   %.2f   %.1f   %.0f" % (x0, 8, t / 8, c.mean().item(), (c * c).mean().item(), (c > 20).double().mean().item(),
                                                                  mm.mean().item(), (mm ** 2).mean().item(), (mm ** 3).mean().item()))
A = np.array(rows)
for cols, name in (((1, 3), "a + c m^2"), ((1, 2, 3), "a + b m + c m^2"), ((1, 4), "a + d m^3"), ((1, 3, 4), "a + c m^2 + d m^3")):
    coef, res, *_ = np.linalg.lstsq(A[:, cols], A[:, 0], rcond=None)
    fit = A[:, cols] @ coef
    print("fit ms/plane = %s: coefficients %s, max rel residual %.3f" % (name, np.array2string(coef, precision=6), np.abs(fit / A[:, 0] - 1).max()))
