#!/usr/bin/env python3
"""Run the bounds-audited twin of a variant (CBET_LIB_PATH = build_alt/libcbet_<name>_audit.so) over a workload that
exercises every write-back path -- 64^3 with all 60 beams, a ragged 33x20x27 grid, a 9x7x13 one -- and print the number of
out-of-range accesses it ATTEMPTED (the audited build counts and skips them).  Exit code 1 when there is any: the plain
build of the same code must then not be run.  Results need not be right (timing variants drop work on purpose)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402,F401
from cbet_raytracing_3d_amd import api   # noqa: E402
from cbet_raytracing_3d_amd.tracer import RayTracer   # noqa: E402

r, ne, te = api.load_s83177()
bn = api.omega60_beam_norm()
total = 0
for nx, ny, nz, rpz, beams in [(64, 64, 64, 4, list(range(60))), (33, 20, 27, 3, [7, 22, 37, 52]), (9, 7, 13, 5, [10, 50])]:
    p = api.default_params(nx, nbeams=len(beams), rays_per_zone=rpz)
    p.ny, p.nz = ny, nz
    tr = RayTracer(p, r, ne, te, beam_norm=np.ascontiguousarray(bn[beams]))
    # dense rows (the reference's layout), rows padded to whole 64-byte lines (what every timed run uses: SweepPipeline) and an
    # odd pitch: the write-back paths address the grid through its strides, and a variant wrong only with a pitch must not
    # pass here and be timed un-audited
    for pitch in (None, True, nz + 2 + 5):
        e = tr.new_grid(zpitch=pitch) if pitch else tr.new_grid()
        tr.launch(e, stats=True)
        torch.cuda.synchronize()
        v = api.debug_bounds_violations(reset=True)
        print("audit %dx%dx%d rpz %d beams %d row pitch %d: %d out-of-range accesses attempted, edep_sum %.10e" % (
            nx, ny, nz, rpz, len(beams), e.shape[2], v, float(e.sum().item())))
        total += v
    tr.close()
sys.exit(1 if total else 0)
