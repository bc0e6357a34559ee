"""Where do the idle lane-steps of the 256^3 pass come from?  CPU, oracle ray paths of one beam (the plasma is spherical
and every beam points at its centre): the step count of every ray of the launch list, hence per bundle the lane-steps it
wastes (64 x longest - sum), by launch radius; and the same for the list with the rim's rays packed (cbet_params.rim_merge).
usage: python scripts/lane_waste.py [n=256]"""
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_inputs  # noqa: E402
from cbet_raytracing_3d_amd import api  # noqa: E402
from oracle import cbet_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bn, r, ne, te = load_inputs()
cfg = O.default_config(n)
p0 = api.default_params(n, rim_merge=0)
d = api.derive(p0)
live0 = api.live_ray_list(p0).reshape(-1, 64)


def lengths(i):
    return [len(O.ray_path(cfg, bn, r, ne, te, 0, int(q))) if q >= 0 else 0 for q in live0[i]]


def report(name, live, L):
    mx = L.max(1)
    waste = 64 * mx - L.sum(1)
    rpz, zx = p0.rays_per_zone, d.zones_spanned
    tile, rem = live // (rpz * rpz), live % (rpz * rpz)
    rx, ry = (tile % zx) * rpz + rem % rpz, (tile // zx) * rpz + rem // rpz
    cnt = np.maximum((live >= 0).sum(1), 1)
    rad = np.hypot(np.where(live >= 0, rx, 0).sum(1) / cnt - d.nrays_x / 2, np.where(live >= 0, ry, 0).sum(1) / cnt - d.nrays_y / 2)
    order = np.argsort(-waste)
    print("%s: %d bundles (%d with holes), lane utilisation %.4f; the worst 10 %% of the bundles carry %.0f %% of the idle lane-steps"
          % (name, len(live), int(((live >= 0).sum(1) < 64).sum()), L.sum() / (64.0 * mx.sum()), 100.0 * waste[order[:len(live) // 10]].sum() / waste.sum()))
    edges = np.linspace(0, rad.max() + 1, 13)
    for a, b in zip(edges[:-1], edges[1:]):
        m = (rad >= a) & (rad < b)
        if m.sum():
            print("    launch radius %3.0f-%3.0f rays: %4d bundles, longest ray %4.0f steps on average, utilisation %.3f, %4.1f %% of the idle lane-steps"
                  % (a, b, m.sum(), mx[m].mean(), L[m].sum() / (64.0 * mx[m].sum()), 100.0 * waste[m].sum() / waste.sum()))


if __name__ == "__main__":
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        L0 = np.array(pool.map(lengths, range(len(live0)), chunksize=8))
    report("one 8x8 patch per bundle (rim_merge = 0)", live0, L0)
    steps = {int(i): int(s) for i, s in zip(live0.ravel(), L0.ravel()) if i >= 0}
    for w in (4,):
        live = api.live_ray_list(api.default_params(n, rim_merge=w)).reshape(-1, 64)
        report("rim_merge = %d" % w, live, np.array([[steps[int(i)] if i >= 0 else 0 for i in b] for b in live]))
