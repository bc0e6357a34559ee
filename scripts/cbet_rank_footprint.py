"""BASELINE config 5's per-rank footprint, for real: ONE rank's arrays of the slab-owned CBET loop at 512^3 / 60 beams / 8 ranks
allocated on one GPU, its field passes and its slab gain update run on them.

The rank's own beams are traced for real (first the four-component, gain-free pass; then the energy-field pass with a gain).
The other ranks' beams over its slab -- what exchange 1 would deliver -- are stand-ins: the rank's own beams' fields over the
same planes, dealt round the 60 beam slots (every OMEGA beam crosses the central planes, so the number of beams per node is
what a real run has there to within the beams' individual footprints).  Reports the device memory in use (hipMemGetInfo
through torch) against cbet_cbet_slab_workspace_bytes_parts + the context's tables, and the two kernel times.

usage: python scripts/cbet_rank_footprint.py [W=8] [n=512] [rank=W//2-1]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api  # noqa: E402
from cbet_raytracing_3d_amd import tracer as T  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rank = int(sys.argv[3]) if len(sys.argv) > 3 else W // 2 - 1
nb = 60
GB = 1e9


def in_use():
    free, total = torch.cuda.mem_get_info()
    return total - free


torch.cuda.init()
torch.zeros(1, device="cuda")
base = in_use()                                    # the HIP context, torch's own pools
r, ne, te = api.load_s83177()
p = api.default_params(n, nbeams=nb)
tr = T.RayTracer(p, r, ne, te)
gp = api.default_gain_params()
X, Y, Z = tr.grid_shape
plane = Y * Z
tables = 8 * n ** 3 * (2 + 4)                     # ne3d + kappa3d + the 32-byte step records
after_ctx = in_use()
print("grid %d^3, %d beams, rank %d of %d" % (n, nb, rank, W))
print("context (node tables, step records, launch lists): %.2f GB in use (tables alone: %.2f GB)" % ((after_ctx - base) / GB, tables / GB))
beams, slabs = T._parts(nb, W), T._parts(X, W)
(b0, b1), (x0, x1) = beams[rank], slabs[rank]
nbr, xr = b1 - b0, x1 - x0
want = api.cbet_slab_workspace_bytes_parts(p, nbr, xr, 0)
eng = T._DeviceCbetEngine(tr, tr.new_grid(), gp)
eng.begin_beams(b0, b1)
eng.begin_slab([(x0, x1)])
torch.cuda.synchronize()
after_arrays = in_use()
print("rank arrays: beams [%d,%d) over the whole grid + all beams over planes [%d,%d): formula %.2f GB, engine holds %.2f GB, device memory grew by %.2f GB"
      % (b0, b1, x0, x1, want / GB, eng.slab_bytes() / GB, (after_arrays - after_ctx) / GB))


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


cur = torch.cuda.current_stream()
t_first = timed(lambda: cur.wait_event(eng.trace_group(0, nbr, False, True)))
c = tr.counters(reset=True)
print("first field pass (four components, no gain) of %d beams: %.1f ms, %.3e ray-steps" % (nbr, t_first, c.ray_steps))
# the peers' beams over my slab: stand-ins dealt from my own beams
for q in range(nb):
    if not (b0 <= q < b1):
        eng.slab_fields[0][:, q].copy_(eng.own_fields[:, (q * 7 + 3) % nbr, x0:x1])
for i in range(nbr):
    eng.slab_fields[0][:, b0 + i].copy_(eng.own_fields[:, i, x0:x1])
keep_e = eng.slab_fields[0][0].clone()                # the raw energy field (the update normalises in place)
t_up0 = timed(lambda: eng.update_gain_slab(False))
print("slab gain update, directions built (%d beams x %d planes): %.1f ms" % (nb, xr, t_up0))
for i in range(nbr):                               # my beams' gain over my own slab; elsewhere it stays zero (the peers' part)
    eng.gain_own[i, x0:x1].copy_(eng.gain_slab[0][b0 + i])
t_field = min(timed(lambda: cur.wait_event(eng.trace_group(0, nbr, True, False))) for _ in range(2))
c = tr.counters(reset=True)
eng.slab_fields[0][0].copy_(keep_e)
t_up = timed(lambda: eng.update_gain_slab(True))
peak = in_use()
print("energy-field pass of %d beams with gain: %.1f ms (%.3e ray-steps/s); slab gain update, directions frozen: %.1f ms" % (nbr, t_field, c.ray_steps / 2 / (t_field * 1e-3), t_up))
extra = keep_e.numel() * 8
print("device memory in use at the end: %.2f GB = context %.2f + rank arrays %.2f + this script's copy of the energy slab %.2f + %.2f other"
      % ((peak - base) / GB, (after_ctx - base) / GB, (after_arrays - after_ctx) / GB, extra / GB, (peak - after_arrays - extra) / GB))
ratio = (after_arrays - after_ctx) / want
print("rank arrays on the device / cbet_cbet_slab_workspace_bytes_parts = %.4f  (%s within 5 %%)" % (ratio, "is" if abs(ratio - 1) < 0.05 else "NOT"))
print("all of it against one MI355X: %.1f GB of 288 GB" % ((after_arrays - base) / GB))
