import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(256), r, ne, te)
for pitch in (None, True):
    e = tr.new_grid(zpitch=pitch) if pitch else tr.new_grid()
    tr.launch(e); torch.cuda.synchronize()
    print("256^3 pitch %d: violations %d edep_sum %.10e" % (e.shape[2], api.debug_bounds_violations(reset=True), float(e.sum().item())), flush=True)
