"""CPU model of a launch's ramp and drain: how long does a rank's 1/K share of the 256^3 sweep take under
different ORDERS of the ray bundles?  Per-bundle step counts come from oracle ray paths of one beam (the plasma is
spherical and every beam points at its centre: all beams have the same distribution); the machine is 256 CUs x 14
waves; a wave-step takes tau(n) microseconds when n waves are resident (measured: 0.456 alone, ~1.76 at full load,
scripts/launch_size_curve.py).  usage: python scripts/tail_model.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_inputs  # noqa: E402
from cbet_raytracing_3d_amd import api  # noqa: E402
from oracle import cbet_oracle as O  # noqa: E402

n = 256
cache = "/tmp/bundle_steps_%d.npy" % n
if os.path.exists(cache):
    steps = np.load(cache)
else:
    bn, r, ne, te = load_inputs()
    cfg = O.default_config(n)
    live = api.live_ray_list(api.default_params(n)).reshape(-1, 64)
    steps = np.zeros(len(live), dtype=np.int64)
    for i, ids in enumerate(live):
        steps[i] = max(len(O.ray_path(cfg, bn, r, ne, te, 0, int(q))) for q in ids if q >= 0)
    np.save(cache, steps)
B = len(steps)
print("bundles per beam %d, wave-steps per beam %d, longest %d, median %d, shortest %d" % (B, steps.sum(), steps.max(), np.median(steps), steps.min()))
SLOTS = 3584
TAU1, TAUF = 0.456, 1.76


def tau(nres):
    return TAU1 + (TAUF - TAU1) * (nres / SLOTS)


def simulate(order_steps, dt=2.0):
    """order_steps: wave-steps of the bundles in dispatch order -> microseconds"""
    q = list(order_steps)
    nq = len(q)
    head = 0
    rem = np.zeros(0)
    t = 0.0
    while head < nq or len(rem):
        free = SLOTS - len(rem)
        if free > 0 and head < nq:
            take = min(free, nq - head)
            rem = np.concatenate([rem, np.array(q[head:head + take], dtype=float)])
            head += take
        rate = dt / tau(len(rem))
        rem = rem - rate
        rem = rem[rem > 0]
        t += dt
    return t


def share(K, rank, nbeams=60):
    total = nbeams * B
    lo, hi = total * rank // K, total * (rank + 1) // K
    return [(g // B, g % B) for g in range(lo, hi)]      # (beam, patch): patches are longest-first inside a beam


for K in (1, 2, 4, 8):
    items = share(K, K // 2)
    base = [steps[p] for _, p in items]
    res = {"beam-major (shipped)": simulate(base)}
    res["global longest-first"] = simulate(sorted(base, reverse=True))
    for C in (2, 4, 8):
        key = sorted(range(len(items)), key=lambda i: (items[i][1] * C // B, items[i][0], items[i][1]))
        res["%d length classes, beam-major inside" % C] = simulate([base[i] for i in key])
    ideal = sum(base) * TAUF / SLOTS
    print("K=%d: %d bundles, ideal (full occupancy throughout) %.2f ms" % (K, len(items), ideal / 1e3))
    for k, v in res.items():
        print("      %-40s %.2f ms" % (k, v / 1e3))
