"""Does the SHAPE of a ray bundle matter?  The shipped bundle is an 8 x 8-ray patch of the beam cross-section (2 x 2 launch
zones).  A ray's path depends mostly on its impact parameter, so rays at the same radius stay together and rays at
different radii fan out after the turning point (what the second deposit box and the window misses are about) and end at
different steps (idle lanes).  Here the same rays are regrouped into 4 x 16-ray patches with the long side along the
TANGENT of the beam cross-section (per sector), along the radius, or fixed, handed to the context with
cbet_context_set_launch_list, and the 256^3 pass is timed against the 8 x 8 list (all without the rim packing).
Lanes are dealt so that the rays of a launch zone still differ in the lane bits 0, 1, 3 that pick the corner order.
usage: python scripts/patch_shapes.py [n=256]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r, ne, te = api.load_s83177()
p = api.default_params(n, rim_merge=0, window_stats=1)
d = api.derive(p)
base = api.live_ray_list(p).reshape(-1, 64)
rpz, zx = p.rays_per_zone, d.zones_spanned
ids = base[base >= 0]
tile, rem = ids // (rpz * rpz), ids % (rpz * rpz)
rx, ry = (tile % zx) * rpz + rem % rpz, (tile // zx) * rpz + rem // rpz
cx, cy = 0.5 * (d.nrays_x - 1), 0.5 * (d.nrays_y - 1)


def lane_of(u, v, w, h):
    """lane of the ray at column u, row v of a w x h patch: bits 0, 1 = column inside the launch zone, bit 3 = row parity"""
    if (w, h) == (8, 8):
        return 8 * v + u
    if (w, h) == (4, 16):
        return (u & 3) | ((v & 1) << 3) | (((v >> 1) & 1) << 2) | (((v >> 2) & 3) << 4)
    if (w, h) == (16, 4):
        return (u & 3) | ((v & 1) << 3) | (((v >> 1) & 1) << 2) | (((u >> 2) & 3) << 4)
    raise ValueError


def build(shape_of):
    """shape_of(rx, ry) -> (w, h) per ray (constant inside every 16 x 16 block so that the patches tile it)"""
    groups = {}
    for q, x, y in zip(ids, rx, ry):
        w, h = shape_of(16 * (x // 16) + 8, 16 * (y // 16) + 8)
        groups.setdefault((w, h, x // w, y // h), []).append((lane_of(x % w, y % h, w, h), q))
    out, key = [], []
    for (w, h, px, py), rays in groups.items():
        b = -np.ones(64, dtype=np.int32)
        for l, q in rays:
            b[l] = q
        out.append(b)
        key.append(-((px * w + w / 2 - cx) ** 2 + (py * h + h / 2 - cy) ** 2))
    order = np.argsort(key, kind="stable")          # longest rays (largest radius) first, like the default list
    return np.stack(out)[order]


tangential = lambda x, y: (4, 16) if abs(x - cx) > abs(y - cy) else (16, 4)
radial = lambda x, y: (16, 4) if abs(x - cx) > abs(y - cy) else (4, 16)
lists = {"8x8 (shipped shape)": build(lambda x, y: (8, 8)), "4x16 long side tangential": build(tangential),
         "4x16 long side radial": build(radial), "4x16 everywhere": build(lambda x, y: (4, 16)), "16x4 everywhere": build(lambda x, y: (16, 4))}
tr = RayTracer(p, r, ne, te)
e = tr.new_grid()
want = None
for rep in range(3):
    for name, lst in lists.items():
        tr.ctx.set_launch_list(lst.ravel())
        ts = []
        for k in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e.zero_(); tr.counters(reset=True); a.record(); tr.launch(e); b.record(); torch.cuda.synchronize()
            if k > 1:
                ts.append(a.elapsed_time(b))
        c = tr.counters(reset=True)
        s = float(e.sum().item())
        want = s if want is None else want
        print("%-28s %5d bundles  %.3f ms (min %.3f)  misses %.3f %%  atomics/step %.4f  lane util %.4f  box B live %.3f  moves/wave-step %.3f  edep_sum %s" % (
            name, len(lst), sum(ts) / len(ts), min(ts), 100.0 * c.lds_evictions / c.ray_steps, c.global_atomics / c.ray_steps,
            c.ray_steps / (64.0 * c.wave_steps), c.wave_steps_wide / c.wave_steps, c.slabs_retired / c.wave_steps,
            "ok" if abs(s / want - 1) < 1e-12 else "DIFFERS %.12e" % s), flush=True)
