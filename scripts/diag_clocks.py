"""Where a wave-step's waiting goes, measured with the shader clock inside the kernel: run the 256^3 trace with a
-DCBET_DIAG_CLOCKS build of the library (scripts/variants/diag_clocks.*; CBET_LIB_PATH must point at it).  That build
stamps s_memtime around the record wait and around the window-shift path (box follow + write-back) of every wave-step
and reports the sums through four of the counter slots.  The stamps cost two scalar-memory reads and one lgkmcnt(0) wait
per stamped stretch, so the absolute time of this build is not the shipped kernel's; the SPLIT is what is read.
usage: CBET_LIB_PATH=build_alt/libcbet_diag_clocks.so python scripts/diag_clocks.py [n=256]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cbet_raytracing_3d_amd import api                      # noqa: E402
from cbet_raytracing_3d_amd.tracer import RayTracer         # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if "diag" not in os.environ.get("CBET_LIB_PATH", ""):
    raise SystemExit("set CBET_LIB_PATH to the -DCBET_DIAG_CLOCKS build")
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n), r, ne, te)
e = tr.new_grid(zpitch=True)
for k in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e.zero_()
    tr.counters(reset=True)
    a.record()
    tr.launch(e)
    b.record()
    torch.cuda.synchronize()
c = tr.counters(reset=True)
ws = float(c.wave_steps)
wait, shift, nshift, total = float(c.global_atomics), float(c.lds_evictions), float(c.wave_steps_wide), float(c.slabs_retired)
print("diag build: launch %.2f ms, %d ray-steps, %.4g wave-steps" % (a.elapsed_time(b), c.ray_steps, ws))
print("per wave-step (shader clocks): wave life %.0f | record wait %.0f (%.1f %%) | window-shift path %.0f (%.1f %%)"
      % (total / ws, wait / ws, 100 * wait / total, shift / ws, 100 * shift / total))
print("window-shift path: entered in %.1f %% of the wave-steps, %.0f clocks per entry" % (100 * nshift / ws, shift / max(1.0, nshift)))
ret, nret = float(c.rays_traced), float(c.wave_steps_miss)
print("  of which write-backs (LDS reads, their waits, atomics): %.0f clocks per wave-step (%.1f %%), %.2f per shift entry, %.0f clocks each"
      % (ret / ws, 100 * ret / total, nret / max(1.0, nshift), ret / max(1.0, nret)))
print("everything else: %.0f clocks per wave-step (%.1f %%)" % ((total - wait - shift) / ws, 100 * (total - wait - shift) / total))
