"""CPU study (oracle ray paths, 256^3): LDS cycles of the deposit's ds_add_f64 for candidate TILE LAYOUTS and lane
mappings, under the bank model measured with scripts/ubench/lds_pattern_cost.hip on MI355X:

    one ds_add_f64 = 8.3 + CB x sum over the four 16-lane groups of (lanes on the group's busiest bank - 1),
    bank = (slot in doubles) mod 16     (same address or not makes little difference: 3.0 vs 2.8 in the micro-benchmark)

usage: python scripts/deposit_layouts.py [--bundles 10] [--pads 137/17,148/18] [--accumulate]

--accumulate scores the shipped kernel's regime instead: a lane's deposits are summed in registers while its eight nodes stay
the same and one ds_add_f64 carries only the lanes whose nodes just changed (the sums of the nodes they leave).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_inputs  # noqa: E402
from oracle import cbet_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--bundles", type=int, default=10)
ap.add_argument("--pads", default="", help="extra padded layouts to score: XS/YS pairs, e.g. 141/17,143/18")
ap.add_argument("--accumulate", action="store_true", help="deposits summed in registers until the ray's nodes change")
ap.add_argument("--xorsep", action="store_true", help="score every separable xor swizzle z ^ (a x & 15) ^ (b y & 15) (shipped mapping only)")
args = ap.parse_args()
CB = 2.0
bn, r, ne, te = load_inputs()
cfg = O.default_config(args.n)
d = O.derive(cfg)
rpz, zones = 4, d.zones_spanned


def ray_id(rx, ry):
    return ((ry // rpz) * zones + rx // rpz) * rpz * rpz + (ry % rpz) * rpz + rx % rpz


_cache = {}


def ray_path(beam, rx, ry):
    key = (beam, rx, ry)
    if key not in _cache:
        if not (0 <= rx < d.nrays_x and 0 <= ry < d.nrays_y):
            _cache[key] = np.zeros((0, 3), dtype=np.int32)
        else:
            _cache[key] = O.ray_path(cfg, bn, r, ne, te, beam, ray_id(rx, ry))[:, 3:6].astype(np.int32)
    return _cache[key]


def b(v, i):
    return (v >> i) & 1


# lane -> (x, y) inside the 8x8 patch, lane -> corner code (3 bits: x, y, z flips)
MAPPINGS = {
    "rowmajor code=x0,x1,y0 (shipped)": (lambda l: (l & 7, l >> 3), lambda l, x, y: (b(x, 0), b(x, 1), b(y, 0))),
    # the 16 lanes of a 16-lane group = the 16 rays of ONE launch zone (4 x 4 rays): lane bits 0-1 ray x, 2-3 ray y, 4 zone x, 5 zone y
    "zonemajor code=x0,x1,y0": (lambda l: (((l >> 4) & 1) * 4 + (l & 3), ((l >> 5) & 1) * 4 + ((l >> 2) & 3)), lambda l, x, y: (b(x, 0), b(x, 1), b(y, 0))),
    "zonemajor code=x0,y0,x1^y1": (lambda l: (((l >> 4) & 1) * 4 + (l & 3), ((l >> 5) & 1) * 4 + ((l >> 2) & 3)), lambda l, x, y: (b(x, 0), b(y, 0), b(x, 1) ^ b(y, 1))),
    # half zones: a group = 2 x 4 rays of each of the two zones side by side is the shipped row-major; 4 x 2 instead:
    "halfzone-y (group = 4 x 2 rays of two zones stacked in y)": (lambda l: (((l >> 5) & 1) * 4 + (l & 3), ((l >> 3) & 1) * 4 + ((l >> 4) & 1) * 2 + ((l >> 2) & 1)), lambda l, x, y: (b(x, 0), b(x, 1), b(y, 0))),
    "rowmajor nocode": (lambda l: (l & 7, l >> 3), lambda l, x, y: (0, 0, 0)),
    "rowmajor code=x0,y0,x1": (lambda l: (l & 7, l >> 3), lambda l, x, y: (b(x, 0), b(y, 0), b(x, 1))),
    "rowmajor code=x0,y0,x1^y1": (lambda l: (l & 7, l >> 3), lambda l, x, y: (b(x, 0), b(y, 0), b(x, 1) ^ b(y, 1))),
}

# slot(X, Y, Z) -> bank (mod 16 of the slot in doubles); X, Y, Z haloed node indices
LAYOUTS = {
    "pad 148/18 (shipped)": lambda X, Y, Z: ((X & 7) * 148 + (Y & 7) * 18 + (Z & 15)) & 15,
    "pad 140/17 (round 2)": lambda X, Y, Z: ((X & 7) * 140 + (Y & 7) * 17 + (Z & 15)) & 15,
    "pad 151/19": lambda X, Y, Z: ((X & 7) * 151 + (Y & 7) * 19 + (Z & 15)) & 15,
    "pad 153/19": lambda X, Y, Z: ((X & 7) * 153 + (Y & 7) * 19 + (Z & 15)) & 15,
    "pad 149/19 (x=5,y=3)": lambda X, Y, Z: ((X & 7) * 149 + (Y & 7) * 19 + (Z & 15)) & 15,
    "dense xor f=x+3y": lambda X, Y, Z: ((Z & 15) ^ (((X & 7) + 3 * (Y & 7)) & 15)),
    "dense xor f=3x+5y... ": lambda X, Y, Z: ((Z & 15) ^ ((3 * (X & 7) + 5 * (Y & 7)) & 15)),
    "dense xor f=7x+3y": lambda X, Y, Z: ((Z & 15) ^ ((7 * (X & 7) + 3 * (Y & 7)) & 15)),
    "dense xor f=(2x)^(5y)": lambda X, Y, Z: ((Z & 15) ^ ((2 * (X & 7)) ^ ((5 * (Y & 7)) & 15))),
    "dense xor f=x^(2y) ": lambda X, Y, Z: ((Z & 15) ^ ((X & 7) ^ (2 * (Y & 7)))),
    "dense xor f=(2x+1)... x*2^y": lambda X, Y, Z: ((Z & 15) ^ ((2 * (X & 7)) ^ (Y & 7))),
    "dense add f=3x+7y (rotate z)": lambda X, Y, Z: (((Z & 15) + 7 * (X & 7) + 3 * (Y & 7)) & 15),     # the CBET kernels' box A
    "dense xor 4x ^ 2y (shipped, plain trace)": lambda X, Y, Z: ((Z & 15) ^ ((4 * (X & 7)) & 15) ^ (2 * (Y & 7))),
    "dense add f=5x+3y": lambda X, Y, Z: (((Z & 15) + 5 * (X & 7) + 3 * (Y & 7)) & 15),
    "dense add f=4x+2y": lambda X, Y, Z: (((Z & 15) + 4 * (X & 7) + 2 * (Y & 7)) & 15),
    "dense add f=8x+4y": lambda X, Y, Z: (((Z & 15) + 8 * (X & 7) + 4 * (Y & 7)) & 15),
    "dense add f=2x+4y": lambda X, Y, Z: (((Z & 15) + 2 * (X & 7) + 4 * (Y & 7)) & 15),
    "dense add f=4x+6y": lambda X, Y, Z: (((Z & 15) + 4 * (X & 7) + 6 * (Y & 7)) & 15),
    "dense add f=6x+2y": lambda X, Y, Z: (((Z & 15) + 6 * (X & 7) + 2 * (Y & 7)) & 15),
    "dense (no swizzle)": lambda X, Y, Z: (Z & 15),
    "ideal (all distinct banks)": None,
}


for _pair in [q for q in args.pads.split(",") if q]:
    _xs, _ys = (int(v) for v in _pair.split("/"))
    assert _xs >= 7 * _ys + 16, "rows overlap"
    LAYOUTS["pad %d/%d" % (_xs, _ys)] = (lambda xs, ys: (lambda X, Y, Z: ((X & 7) * xs + (Y & 7) * ys + (Z & 15)) & 15))(_xs, _ys)
if args.xorsep:
    LAYOUTS = {k: v for k, v in LAYOUTS.items() if "rotate z" in k or "ideal" in k}
    for _a in range(16):
        for _b in range(16):
            LAYOUTS["xor sep a=%d b=%d" % (_a, _b)] = (lambda a_, b_: (lambda X, Y, Z: (Z & 15) ^ ((a_ * (X & 7)) & 15) ^ ((b_ * (Y & 7)) & 15)))(_a, _b)
if args.pads or args.xorsep:
    MAPPINGS = {k: v for k, v in MAPPINGS.items() if "shipped" in k}


def bundle_instructions(beam, bx, by, xy, code):
    lane_xy = [xy(l) for l in range(64)]
    codes = np.array([code(l, *lane_xy[l]) for l in range(64)])
    paths = [ray_path(beam, bx + x, by + y) for x, y in lane_xy]
    T = max(len(p) for p in paths)
    prev, had = np.zeros((64, 3), dtype=np.int64), np.zeros(64, dtype=bool)
    for t in range(T + (1 if args.accumulate else 0)):
        low = np.zeros((64, 3), dtype=np.int64)
        act = np.zeros(64, dtype=bool)
        for l in range(64):
            if len(paths[l]) > t:
                low[l] = paths[l][t]
                act[l] = True
        if args.accumulate:   # the lanes whose nodes change (or whose ray ended) flush the sums of the nodes they leave
            fl = had & ((low != prev).any(axis=1) | ~act)
            cur_low, cur_had = low.copy(), act.copy()
            low, act = prev, fl
            prev, had = cur_low, cur_had
            if not fl.any():
                yield None
                continue
        for c in range(8):
            cx, cy, cz = c & 1, (c >> 2) & 1, (c >> 1) & 1
            yield (low[:, 0] + (cx ^ codes[:, 0] ^ 1), low[:, 1] + (cy ^ codes[:, 1] ^ 1), low[:, 2] + (cz ^ codes[:, 2] ^ 1), act)


def cost(bank, act):
    c = 8.3
    for g in range(4):
        s = bank[16 * g:16 * g + 16][act[16 * g:16 * g + 16]]
        if len(s):
            c += CB * (np.bincount(s, minlength=16).max() - 1)
    return c


rng = np.random.default_rng(3)
px, py = (d.nrays_x + 7) // 8, (d.nrays_y + 7) // 8
samples = []
while len(samples) < args.bundles:
    bx, by = int(rng.integers(px)), int(rng.integers(py))
    cx, cy = bx * 8 + 4 - d.nrays_x / 2, by * 8 + 4 - d.nrays_y / 2
    if cx * cx + cy * cy <= (d.nrays_x / 2 - 10) ** 2:
        samples.append((int(rng.integers(60)), bx, by))
for mname, (xy, code) in MAPPINGS.items():
    tot = {k: 0.0 for k in LAYOUTS}
    n = 0
    for beam, bx, by in samples:
        for ins in bundle_instructions(beam, bx * 8, by * 8, xy, code):
            n += 1
            if ins is None:       # a wave-step without a flush: eight instruction slots that cost nothing
                n += 7
                continue
            X, Y, Z, act = ins
            for lname, f in LAYOUTS.items():
                tot[lname] += 8.3 if f is None else cost(f(X, Y, Z), act)
    print(mname)
    for lname in LAYOUTS:
        print("    %-34s %5.1f cycles per ds_add_f64  (%5.1f per wave-step)" % (lname, tot[lname] / n, 8 * tot[lname] / n))
