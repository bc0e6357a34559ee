"""CPU study (oracle ray paths, 256^3): what candidate deposit schemes would ask of the LDS per wave-step.

For sampled ray bundles, replays every ray-step's eight deposit nodes under a lane mapping (which ray of the patch a
lane carries, which corner order it uses) and reports the ds_add_f64 cost per wave-step under the cost model measured
with scripts/ubench/lds_atomic2.hip on MI355X:

    one ds_add_f64 = 8.3 CU cycles + 3 x sum over the four 16-lane groups (lanes 0-15, 16-31, ...) of
                     (lanes on the group's busiest address - 1)  [+ ~0.9 per adjacent pair of groups sharing an address]

i.e. address conflicts only cost inside a 16-lane group.  `--export FILE` writes sampled instruction patterns
(int32 [n][64] slots, -1 = inactive) for scripts/ubench/lds_pattern_cost.hip, which measures them on the GPU.

usage: python scripts/deposit_schemes.py [--n 256] [--bundles 12] [--export patterns.bin]
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
ap.add_argument("--bundles", type=int, default=12)
ap.add_argument("--export", default=None)
ap.add_argument("--per-scheme", type=int, default=160)
args = ap.parse_args()
n = args.n
bn, r, ne, te = load_inputs()
cfg = O.default_config(n)
d = O.derive(cfg)
rpz = 4
zones = d.zones_spanned


def ray_id(rx, ry):
    tile = (ry // rpz) * zones + rx // rpz
    return tile * rpz * rpz + (ry % rpz) * rpz + rx % rpz


_cache = {}


def ray_path(beam, rx, ry):
    key = (beam, rx, ry)
    if key not in _cache:
        if rx >= d.nrays_x or ry >= d.nrays_y or rx < 0 or ry < 0:
            _cache[key] = np.zeros((0, 8))
        else:
            _cache[key] = O.ray_path(cfg, bn, r, ne, te, beam, ray_id(rx, ry))[:, 3:6].astype(np.int32)
    return _cache[key]


def bits(v, i):
    return (v >> i) & 1


class Scheme:
    """lane -> (x, y) offset inside the block, lane -> 3-bit corner code, tile strides, options"""

    def __init__(self, name, xy, code, XS=140, YS=17, span=8, regs=False):
        self.name, self.xy, self.code, self.XS, self.YS, self.span, self.regs = name, xy, code, XS, YS, span, regs


def rowmajor(l):
    return l & 7, l >> 3


def code_now(l, x, y):
    return bits(x, 0), bits(x, 1), bits(y, 0)


def lanes_from(perm_bits):
    """perm_bits: for lane bit i, which ray-coordinate bit ('x0','y2',...) it carries"""
    def f(l):
        x = y = 0
        for i, b in enumerate(perm_bits):
            if bits(l, i):
                if b[0] == 'x':
                    x |= 1 << int(b[1])
                else:
                    y |= 1 << int(b[1])
        return x, y
    return f


SCHEMES = [
    Scheme("now", rowmajor, code_now),
    Scheme("now+regs", rowmajor, code_now, regs=True),
    # 16-lane group = the stride-2 sublattice (x0, y0 select the group)
    Scheme("grp=x0y0 code=x1,y1,x2^y2", lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"]), lambda l, x, y: (bits(x, 1), bits(y, 1), bits(x, 2) ^ bits(y, 2))),
    Scheme("grp=x0y0 code=x1,y1,x2", lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"]), lambda l, x, y: (bits(x, 1), bits(y, 1), bits(x, 2))),
    Scheme("grp=x0y0 nocode", lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"]), lambda l, x, y: (0, 0, 0)),
    Scheme("grp=x0y0 code=x1,y1,0", lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"]), lambda l, x, y: (bits(x, 1), bits(y, 1), 0)),
    # 16-lane group = (x1, y1)
    Scheme("grp=x1y1 code=x0,y0,x2^y2", lanes_from(["x0", "y0", "x2", "y2", "x1", "y1"]), lambda l, x, y: (bits(x, 0), bits(y, 0), bits(x, 2) ^ bits(y, 2))),
    # 16-lane group = (x0, x1): one column of each zone pair ... rays x fixed mod 4, all y
    Scheme("grp=x0x1 code=y0,y1,x2", lanes_from(["y0", "y1", "y2", "x2", "x0", "x1"]), lambda l, x, y: (bits(y, 0), bits(y, 1), bits(x, 2))),
    Scheme("grp=x0y0 +regs", lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"]), lambda l, x, y: (bits(x, 1), bits(y, 1), bits(x, 2) ^ bits(y, 2)), regs=True),
    Scheme("stride2 rowmajor", lambda l: (2 * (l & 7), 2 * (l >> 3)), lambda l, x, y: (bits(x, 1), bits(y, 1), bits(x, 2) ^ bits(y, 2)), span=16),
    Scheme("stride2 grp=x1y1 nocode", lambda l: tuple(2 * v for v in lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"])(l)), lambda l, x, y: (0, 0, 0), span=16),
    Scheme("stride2 grp=x1y1 code=grp", lambda l: tuple(2 * v for v in lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"])(l)), lambda l, x, y: (bits(x, 1), bits(y, 1), 0), span=16),
    Scheme("stride2 grp=x1y1 nocode +regs", lambda l: tuple(2 * v for v in lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"])(l)), lambda l, x, y: (0, 0, 0), span=16, regs=True),
    Scheme("stride4 32x32 nocode", lambda l: tuple(4 * v for v in lanes_from(["x0", "y0", "x1", "y1", "x2", "y2"])(l)), lambda l, x, y: (0, 0, 0), span=32),
    Scheme("stride2 grp=x1y1", lambda l: tuple(2 * v for v in lanes_from(["x1", "y1", "x2", "y2", "x0", "y0"])(l)), lambda l, x, y: (bits(x, 2), bits(y, 2), bits(x, 3) ^ bits(y, 3)), span=16),
]


def model_cost(slots):
    """slots: int array [64], -1 inactive"""
    c = 8.3
    groups = []
    for g in range(4):
        s = slots[16 * g:16 * g + 16]
        s = s[s >= 0]
        if len(s) == 0:
            groups.append(set())
            continue
        _, cnt = np.unique(s, return_counts=True)
        c += 3.0 * (cnt.max() - 1)
        groups.append(set(s.tolist()))
    for g in range(3):
        if groups[g] & groups[g + 1]:
            c += 0.9
    return c


def analyse(beam, bx, by, sch, rng, export):
    lane_xy = [sch.xy(l) for l in range(64)]
    codes = np.array([sch.code(l, *lane_xy[l]) for l in range(64)])
    paths = [ray_path(beam, bx + x, by + y) for x, y in lane_xy]
    T = max(len(p) for p in paths)
    if T == 0:
        return []
    out = []
    prev = [None] * 64
    for t in range(T + 1):
        low = np.full((64, 3), -1, dtype=np.int64)
        act = np.zeros(64, dtype=bool)
        nlive = 0
        for l in range(64):
            p = paths[l]
            cube = tuple(p[t]) if len(p) > t else None
            if cube is not None:
                nlive += 1
            if sch.regs:
                if prev[l] is not None and cube != prev[l]:
                    low[l] = prev[l]
                    act[l] = True
            elif cube is not None:
                low[l] = cube
                act[l] = True
            prev[l] = cube
        if nlive == 0 and not act.any():
            break
        if not act.any():
            out.append((0.0, 0, nlive))
            continue
        cyc = 0.0
        for c in range(8):
            cx, cy, cz = c & 1, (c >> 2) & 1, (c >> 1) & 1
            X = low[:, 0] + (cx ^ codes[:, 0] ^ 1)     # first-visited = own (low + 1) unless flipped
            Y = low[:, 1] + (cy ^ codes[:, 1] ^ 1)
            Z = low[:, 2] + (cz ^ codes[:, 2] ^ 1)
            slot = (X & 7) * sch.XS + (Y & 7) * sch.YS + (Z & 15)
            slot = np.where(act, slot, -1)
            cyc += model_cost(slot)
            if export is not None and rng.random() < 0.02:
                export.append(slot.astype(np.int32))
        out.append((cyc, int(act.sum()), nlive))
    return out


rng = np.random.default_rng(3)
px, py = (d.nrays_x + 7) // 8, (d.nrays_y + 7) // 8
samples = []
while len(samples) < args.bundles:
    bx, by = int(rng.integers(px)), int(rng.integers(py))
    cx, cy = bx * 8 + 4 - d.nrays_x / 2, by * 8 + 4 - d.nrays_y / 2
    if cx * cx + cy * cy > (d.nrays_x / 2 - 10) ** 2:
        continue
    samples.append((int(rng.integers(60)), bx, by))
allpat, names = [], []
for sch in SCHEMES:
    rows = []
    exp = [] if args.export else None
    for beam, bx, by in samples:
        if sch.span == 32:
            rows += analyse(beam, (bx // 4) * 32 + (bx & 3), (by // 4) * 32 + (by & 3), sch, rng, exp)
        elif sch.span == 16:   # the four interleaved patches of the 16x16 block share the work: take the one holding this patch
            rows += analyse(beam, (bx // 2) * 16 + (bx & 1), (by // 2) * 16 + (by & 1), sch, rng, exp)
        else:
            rows += analyse(beam, bx * 8, by * 8, sch, rng, exp)
    a = np.array(rows, dtype=float)
    print("%-34s wave-steps %6d  active lanes %5.1f  model LDS cycles per wave-step %6.1f (per ds_add %5.1f)"
          % (sch.name, len(a), a[:, 1].mean(), a[:, 0].mean(), a[:, 0].mean() / 8))
    if exp:
        idx = rng.choice(len(exp), min(args.per_scheme, len(exp)), replace=False)
        for i in idx:
            allpat.append(exp[i])
            names.append("%s|%.1f" % (sch.name, model_cost(exp[i])))
if args.export:
    np.array(allpat, dtype=np.int32).tofile(args.export)
    open(args.export + ".names", "w").write("\n".join(names) + "\n")
    print("exported %d patterns to %s" % (len(allpat), args.export))
