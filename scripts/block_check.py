#!/usr/bin/env python3
"""Parity of CBET_KERNEL_LDS_BLOCK (cbet_trace_block.hip) against the oracle on small and ragged grids, with the library in
CBET_LIB_PATH (run the bounds-audited twin first: it counts out-of-range accesses instead of performing them).
usage: block_check.py [quick|full]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np   # noqa: E402
import torch         # noqa: E402
from conftest import load_inputs, parity_err   # noqa: E402
from cbet_raytracing_3d_amd import api   # noqa: E402
from cbet_raytracing_3d_amd.tracer import RayTracer   # noqa: E402
from oracle import cbet_oracle as O   # noqa: E402

bn, r, ne, te = load_inputs()
cases = [(3, 3, 3, 4, [0, 30]), (9, 7, 13, 5, [10, 50]), (16, 16, 16, 4, [0, 15, 30, 45]), (33, 20, 27, 3, [7, 22, 37, 52]),
         (48, 48, 48, 4, list(range(0, 60, 8))), (64, 64, 64, 4, list(range(0, 60, 3)))]
if len(sys.argv) > 1 and sys.argv[1] == "full":
    cases.append((100, 100, 100, 4, list(range(60))))
audited = True
bad = 0
for nx, ny, nz, rpz, beams in cases:
    p = api.default_params(nx, nbeams=len(beams), rays_per_zone=rpz, kernel_variant=api.KERNEL_LDS_BLOCK)
    p.ny, p.nz = ny, nz
    tr = RayTracer(p, r, ne, te, beam_norm=bn[beams])
    e = tr.new_grid()
    tr.counters(reset=True)
    tr.launch(e)
    torch.cuda.synchronize()
    c = tr.counters(reset=True)
    viol = -1
    if audited:
        try:
            viol = api.debug_bounds_violations(reset=True)
        except api.CbetError:
            audited = False
    cfg = O.default_config(nx, nbeams=len(beams), rays_per_zone=rpz)
    cfg.ny, cfg.nz = ny, nz
    oe, osteps = O.trace(cfg, bn[beams].copy(), r, ne, te, nthreads=8)
    got = e.cpu().numpy()
    err = parity_err(got, oe) if np.abs(oe).max() > 0 else float(np.abs(got).max())
    ok = err < 1e-9 and int(c.ray_steps) == int(osteps) and viol <= 0
    bad += 0 if ok else 1
    print("%-4s %3dx%3dx%3d rpz %d beams %2d: steps %d / %d  err %.2e  violations %d  atomics/step %.3f  miss %.3f%%  wave-steps %d  shifts %d" % (
        "ok" if ok else "BAD", nx, ny, nz, rpz, len(beams), c.ray_steps, osteps, err, viol, c.global_atomics / max(1, c.ray_steps),
        100.0 * c.lds_evictions / max(1, c.ray_steps), c.wave_steps, c.slabs_retired), flush=True)
    tr.close()
sys.exit(1 if bad else 0)
