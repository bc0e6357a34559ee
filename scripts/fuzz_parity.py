"""Randomised parity sweep on the GPU: small random grids (ragged, tiny, face-hugging), beam subsets, rays per
zone, absorption on/off, sharding, beam-resolved grids, kernel variants and window knobs -- every case against
the CPU oracle (SURVEY 8(c) metric <= 1e-9, equal ray-step counts).  usage: python scripts/fuzz_parity.py [cases=40] [seed=1]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_inputs, parity_err
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
from oracle import cbet_oracle as O
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bn, r, ne, te = load_inputs()
worst, bad = 0.0, 0
for case in range(cases):
    nx, ny, nz = (int(v) for v in rng.integers(3, 56, size=3))
    if rng.random() < 0.3: ny = nz = nx
    rpz = int(rng.integers(1, 7))
    nb = int(rng.integers(1, 7))
    beams = sorted(rng.choice(60, size=nb, replace=False).tolist())
    absorb = int(rng.random() < 0.8)
    variant = int(rng.choice([0, 1, 2, 3]))
    kw = {}
    if variant in (0, 3):
        mode = rng.integers(0, 4)
        if mode == 1: kw = dict(lds_two_boxes=0)
        elif mode == 2: kw = dict(lds_two_boxes=0, lds_corner_flip=0, lds_copies_log2=int(rng.integers(0, 3)), lds_prereduce=int(rng.integers(0, 3)))
        elif mode == 3: kw = dict(lds_window_log2=4)
    elif variant == 2 and rng.random() < 0.5:
        kw = dict(lds_window_log2=4)
    wide = int(rng.random() < 0.2)
    shards = int(rng.choice([1, 1, 2, 3]))
    per_beam = bool(rng.random() < 0.3)
    p = api.default_params(nx, nbeams=nb, rays_per_zone=rpz, absorption=absorb, kernel_variant=variant, force_wide_index=wide)
    p.ny, p.nz = ny, nz
    desc = dict(grid=(nx, ny, nz), rpz=rpz, beams=beams, absorption=absorb, variant=variant, knobs=kw, wide=wide, shards=shards, per_beam=per_beam)
    try:
        tr = RayTracer(p, r, ne, te, beam_norm=bn[beams])
    except api.CbetError as exc:
        print("case %d skipped (%s): %s" % (case, exc, desc)); continue
    e = tr.new_grid(per_beam=per_beam)
    tr.counters(reset=True)
    for s in range(shards):
        tr.launch(e, shard_index=s, shard_count=shards, **kw)
    c = tr.counters(reset=True)
    cfg = O.default_config(nx, nbeams=nb, rays_per_zone=rpz, absorption=absorb)
    cfg.ny, cfg.nz = ny, nz
    got = e.cpu().numpy()
    if per_beam:
        errs, osteps = [], 0
        for b in range(nb):
            ob, st = O.trace(cfg, bn[beams].copy(), r, ne, te, beam_lo=b, beam_hi=b + 1, nthreads=8)
            osteps += st
            errs.append(parity_err(got[b], ob) if np.abs(ob).max() > 0 else float(np.abs(got[b]).max()))
        err = max(errs)
    else:
        oe, osteps = O.trace(cfg, bn[beams].copy(), r, ne, te, nthreads=8)
        err = parity_err(got, oe) if np.abs(oe).max() > 0 else float(np.abs(got).max())
    ok = err < 1e-9 and c.ray_steps == osteps
    worst = max(worst, err)
    bad += not ok
    print("case %2d %s err %.2e steps %d/%d %s" % (case, "ok  " if ok else "FAIL", err, c.ray_steps, osteps, "" if ok else desc), flush=True)
    tr.close()
print("cases %d, failures %d, worst err %.2e" % (cases, bad, worst))
sys.exit(1 if bad else 0)
