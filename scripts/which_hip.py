"""Diagnostic: which HIP / HSA / RCCL shared objects are mapped, for both load orders."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r)
order = sys.argv[1]
if order == "torch_first":
    import torch
    from cbet_raytracing_3d_amd import api; api.lib()
else:
    from cbet_raytracing_3d_amd import api; api.lib()
    import torch
print(order, "cuda available:", torch.cuda.is_available())
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("amdhip64", "hsa-runtime", "librccl"))})
for l in libs: print("   ", l)
if torch.cuda.is_available():
    t = torch.zeros(4, device="cuda")
    try:
        c = api.Context(api.default_params(16, nbeams=1), 0); print("    context ok"); c.close()
    except Exception as e:
        print("    context FAILED:", e)
''' % ROOT
for order in ("torch_first", "lib_first"):
    subprocess.run([sys.executable, "-c", CODE, order])
