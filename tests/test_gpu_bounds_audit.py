"""Edge-case geometries through a bounds-audited build of the library.

The audit build (-DCBET_DEBUG_BOUNDS, compiled here with hipcc into a temp dir) range-checks every
grid atomic, node-table gather and LDS accumulate in the kernels, counting and skipping violations.
Tiny, ragged and face-hugging configurations run through it in a subprocess (the library path is
fixed at import) and must (a) attempt no out-of-range access and (b) match the oracle.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from conftest import load_inputs, parity_err
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
from oracle import cbet_oracle as O
bn, r, ne, te = load_inputs()
cases = [  # (nx, ny, nz, rays_per_zone, beams, absorption)
    (3, 3, 3, 4, [0, 30], 1), (4, 5, 3, 2, [3, 41], 1), (8, 8, 8, 1, [0, 1, 2], 1), (9, 7, 13, 5, [10, 50], 1),
    (16, 16, 16, 4, [0, 15, 30, 45], 0), (33, 20, 27, 3, [7, 22, 37, 52], 1), (64, 64, 64, 4, list(range(0, 60, 6)), 1),
]
out = []
for nx, ny, nz, rpz, beams, absorb in cases:
    for variant in (1, 2, 3):
        p = api.default_params(nx, nbeams=len(beams), rays_per_zone=rpz, absorption=absorb, kernel_variant=variant)
        p.ny, p.nz = ny, nz
        tr = RayTracer(p, r, ne, te, beam_norm=bn[beams])
        e = tr.new_grid(); tr.counters(reset=True)
        tr.launch(e)
        c = tr.counters(reset=True)
        viol = api.debug_bounds_violations(reset=True)
        cfg = O.default_config(nx, nbeams=len(beams), rays_per_zone=rpz, absorption=absorb)
        cfg.ny, cfg.nz = ny, nz
        oe, osteps = O.trace(cfg, bn[beams].copy(), r, ne, te, nthreads=8)
        err = parity_err(e.cpu().numpy(), oe) if np.abs(oe).max() > 0 else float(np.abs(e.cpu().numpy()).max())
        out.append(dict(case=[nx, ny, nz, rpz, len(beams), absorb], variant=variant, violations=viol,
                        steps=int(c.ray_steps), osteps=int(osteps), err=err))
        tr.close()
# the CBET hooks through the same audit: fused field pass and deposition pass with a gain field
cbet = []
for nx, ny, nz, rpz, beams in [(3, 3, 3, 4, [0, 30]), (9, 7, 13, 5, [10, 50, 20]), (33, 20, 27, 3, [7, 22, 37, 52])]:
    p = api.default_params(nx, nbeams=len(beams), rays_per_zone=rpz)
    p.ny, p.nz = ny, nz
    tr = RayTracer(p, r, ne, te, beam_norm=bn[beams])
    tr.tabulate()
    gp = api.default_gain_params()
    cfg = O.default_config(nx, nbeams=len(beams), rays_per_zone=rpz)
    cfg.ny, cfg.nz = ny, nz
    ne3d, kap = O.node_tables(cfg, r, ne, te)
    rng = np.random.default_rng(nx)
    gain = rng.uniform(-40.0, 40.0, size=(len(beams), nx + 2, ny + 2, nz + 2))
    d_gain = torch.from_numpy(gain).cuda()
    f = tr.new_fields(); tr.counters(reset=True)
    tr.launch_cbet(f, gp, fields=True, gain=d_gain)
    e = tr.new_grid()
    tr.launch_cbet(e, gp, gain=d_gain)
    c = tr.counters(reset=True)
    viol = api.debug_bounds_violations(reset=True)
    og = O.gain_default()
    of = np.stack([O.trace_cbet(cfg, og, bn[beams].copy(), ne3d, kap, gain=gain, quantity=q, per_beam=True, nthreads=8)[0]
                   for q in (1, 2, 3, 4)])
    oe, osteps, _ = O.trace_cbet(cfg, og, bn[beams].copy(), ne3d, kap, gain=gain, nthreads=8)
    scale = max(np.abs(of).max(), 1e-300)
    cbet.append(dict(case=[nx, ny, nz, rpz, len(beams)], violations=viol, steps=int(c.ray_steps), osteps=2 * int(osteps),
                     err_fields=float(np.abs(f.cpu().numpy() - of).max() / scale),
                     err_edep=parity_err(e.cpu().numpy(), oe) if np.abs(oe).max() > 0 else 0.0))
    tr.close()
print("RESULT " + json.dumps(out))
print("CBET " + json.dumps(cbet))
'''


def test_edge_geometries_in_bounds_audited_build(tmp_path):
    csrc = os.path.join(ROOT, "cbet_raytracing_3d_amd", "csrc")
    lib = str(tmp_path / "libcbet_audit.so")
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC",
                           "-shared", "-DCBET_DEBUG_BOUNDS", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                           "-o", lib] + srcs + ["-lrccl"])
    env = dict(os.environ, CBET_LIB_PATH=lib)
    run = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=env, capture_output=True,
                         text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    line = [l for l in run.stdout.splitlines() if l.startswith("RESULT ")][-1]
    results = json.loads(line[len("RESULT "):])
    assert len(results) == 21
    for res in results:
        assert res["violations"] == 0, res
        assert res["steps"] == res["osteps"], res
        assert res["err"] < 1e-9, res
    cbet = json.loads([l for l in run.stdout.splitlines() if l.startswith("CBET ")][-1][len("CBET "):])
    assert len(cbet) == 3
    for res in cbet:   # the CBET hooks: gain gathers, four-component tiles and flushes
        assert res["violations"] == 0, res
        assert res["steps"] == res["osteps"], res
        assert res["err_fields"] < 1e-9 and res["err_edep"] < 1e-9, res


def test_regular_build_reports_no_audit(tmp_path):
    from cbet_raytracing_3d_amd import api
    with pytest.raises(api.CbetError) as ei:
        api.debug_bounds_violations()
    assert ei.value.code == api.EINVAL
