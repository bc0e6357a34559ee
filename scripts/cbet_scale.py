"""CBET stage at scale: time of the field passes, the normalise + gain kernels and the whole solve.
usage: python scripts/cbet_scale.py [n=256] [nbeams=60] [max passes=12] [solve: only the whole solve (one workspace, no separate
       field / gain arrays: what fits beside it at 512^3)]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
gp = api.default_gain_params(tolerance=1e-4, max_passes=int(sys.argv[3]) if len(sys.argv) > 3 else 12)
solve_only = len(sys.argv) > 4 and sys.argv[4] == "solve"
print("n=%d beams=%d workspace %.1f GB" % (n, nb, api.cbet_workspace_bytes(tr.params) / 1e9), flush=True)

def timed(fn, reps=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

e = tr.new_grid()
tr.launch(e); torch.cuda.synchronize()
tr.counters(reset=True)
t_ref = timed(lambda: tr.launch(e), 3)
steps_ref = tr.counters(reset=True).ray_steps // 3
print("reference pass: %.2f ms (tabulate + trace), %d ray-steps" % (t_ref, steps_ref), flush=True)
if solve_only:
    tr.tabulate()
else:
  fields, gain = tr.new_fields(), tr.new_grid(per_beam=True)
  tr.tabulate()
  print("field pass (4 components, one trace), no gain: %.2f ms" % timed(lambda: tr.launch_cbet(fields, gp, fields=True)), flush=True)
  change = torch.zeros(2, dtype=torch.float64, device="cuda")
  scratch = torch.empty_like(gain)
  t = timed(lambda: tr.gain_field(fields, gain, gp, change, scratch=scratch))
  print("normalise + gain kernels: %.2f ms; K max %.3g 1/cm" % (t, float(gain.abs().max())), flush=True)
  fields.zero_()
  print("field pass, with gain: %.2f ms" % timed(lambda: tr.launch_cbet(fields, gp, fields=True, gain=gain)), flush=True)
  fields[0].zero_()
  print("energy-field pass (every pass after the first), with gain: %.2f ms" % timed(lambda: tr.launch_cbet(fields[0], gp, fields="energy", gain=gain)), flush=True)
  t = timed(lambda: tr.gain_field(fields, gain, gp, change, scratch=scratch, frozen=True))
  print("gain kernel, frozen directions: %.2f ms" % t, flush=True)
  e.zero_()
  print("deposition pass with gain: %.2f ms" % timed(lambda: tr.launch_cbet(e, gp, gain=gain)), flush=True)
  del fields, gain, scratch
torch.cuda.empty_cache()
e.zero_()
ws = torch.empty(api.cbet_workspace_bytes(tr.params) // 8, dtype=torch.float64, device="cuda")
t0 = time.perf_counter()
rep = api.cbet_solve(tr.d_te, tr.d_r, tr.d_ne, e, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                     tr.params, gp, workspace=ws, ctx=tr.ctx, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bg = np.array(rep.beam_gain[:nb])
print("solve: %d passes, converged=%d, change %.2e, imbalance %.2e, %.1f ms total, %d ray-steps traced (%d in the final pass)"
      % (rep.passes, rep.converged, rep.change, rep.imbalance, dt * 1e3, rep.ray_steps, rep.ray_steps_final))
print("  -> %.3g ray-steps/s over the whole solve; per-beam gain / injected: min %.3f max %.3f (units of |gain|/sum edep)"
      % (rep.ray_steps / dt, bg.min() / float(e.sum()) * nb, bg.max() / float(e.sum()) * nb))
