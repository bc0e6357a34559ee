"""Regenerates tests/golden/*.npz|json from the CPU oracle (oracle/cbet_oracle.c).

Run from the repo root:  python tests/golden/make_golden.py
The oracle is first checked against the reference known answers recorded in SURVEY.md 8(c)
(tests/test_oracle_golden.py does the same on every CPU test run); the fixtures written here are
data only: inputs are the committed s83177 profiles + OMEGA-60 table, outputs are planes / sums /
single-ray traces of the oracle's result.
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_inputs  # noqa: E402
from oracle import cbet_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def full_case(n, bn, r, ne, te):
    cfg = O.default_config(n)
    edep, steps, per_beam = O.trace(cfg, bn, r, ne, te, nthreads=os.cpu_count(), want_per_beam=True)
    c = (n + 2) // 2
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "edep.txt")
        nbytes = O.write_text(edep, path)
        md5 = hashlib.md5(open(path, "rb").read()).hexdigest()
    np.savez_compressed(os.path.join(OUT, "planes_%d.npz" % n), yz=edep[c], xz=edep[:, c], xy=edep[:, :, c],
                        face_x0=edep[0], face_z0=edep[:, :, 0])
    per_beam_sum = np.zeros(cfg.nbeams)
    # per-beam sums from separate single-beam traces of a few beams (cheap) -- beams are independent
    for b in (0, 17, 59):
        e1, _ = O.trace(cfg, bn, r, ne, te, beam_lo=b, beam_hi=b + 1, nthreads=os.cpu_count())
        per_beam_sum[b] = e1.sum()
    return {
        "n": n, "ray_steps": steps, "sum": float(edep.sum()), "max": float(edep.max()),
        "nonzero": int(np.count_nonzero(edep)), "negative": int((edep < 0).sum()),
        "text_bytes": nbytes, "text_md5": md5,
        "steps_per_beam": per_beam.tolist(),
        "beam_sums": {str(b): per_beam_sum[b] for b in (0, 17, 59)},
        "cells": {"1,1,1": float(edep[1, 1, 1]), "%d,%d,%d" % (c, c, n - 10): float(edep[c, c, n - 10]),
                  "20,%d,%d" % (c, c): float(edep[20, c, c])},
    }


def main():
    bn, r, ne, te = load_inputs()
    meta = {"cases": [full_case(n, bn, r, ne, te) for n in (64, 100)]}

    # single-ray known answers (positions, cells, increments) incl. a culled ray
    cfg = O.default_config(100)
    rays = {}
    for beam, ray in ((0, 9000), (17, 1775), (42, 12345), (59, 19455), (3, 0)):
        live, lp = O.launch_point(cfg, bn, beam, ray)
        path = O.ray_path(cfg, bn, r, ne, te, beam, ray)
        rays["%d,%d" % (beam, ray)] = {"live": live, "launch": lp.tolist(), "steps": int(len(path)),
                                       "first": path[:3].tolist(), "last": path[-2:].tolist() if len(path) else []}
    meta["rays_100"] = rays

    # interp known answers (launch_ray_XZ.cu:16-63), incl. clamps and exact knots
    xs = [0.0, 1e-9, r[1], r[200], 0.5 * (r[200] + r[201]), r[442], 0.3, 1.0, -1.0, 0.0437]
    meta["interp_ne"] = [[x, O.interp(ne, r, x)] for x in map(float, xs)]
    meta["interp_te"] = [[x, O.interp(te, r, x)] for x in map(float, xs)]

    # config 1 of BASELINE.json: 2 crossing beams, 64^3, uniform plasma, no absorption
    cfg1 = O.default_config(64, nbeams=2, absorption=0)
    d1 = O.derive(cfg1)
    bn2 = bn[[0, 30]].copy()
    ne_u = np.full(443, 0.1 * d1.ncrit)
    te_u = np.full(443, 2000.0)
    e1, s1 = O.trace(cfg1, bn2, r, ne_u, te_u, nthreads=os.cpu_count())
    c = 33
    np.savez_compressed(os.path.join(OUT, "planes_cfg1_64.npz"), yz=e1[c], xz=e1[:, c], xy=e1[:, :, c])
    meta["config1_64"] = {"ray_steps": s1, "sum": float(e1.sum()), "max": float(e1.max()),
                          "nonzero": int(np.count_nonzero(e1)), "ne_over_ncrit": 0.1, "te": 2000.0,
                          "beams": [0, 30]}
    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
