"""The RCCL code paths on a one-GPU box: a one-rank "nccl" process group with every collective forced to run
(tests/helpers/rccl_one_rank_smoke.py), and the native orchestrator's ncclReduceScatter combine on a one-device
communicator (CBET_FORCE_RCCL=1).  RCCL allows one rank per device, so the transport over xGMI itself can only run on
a multi-GPU node; what runs here are the calls, their stream ordering and their results."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_inputs, parity_err

pytestmark = pytest.mark.gpu


def test_torch_rccl_paths_on_a_one_rank_group():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_one_rank_smoke.py")],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert run.returncode == 0 and "RCCL SMOKE PASS" in run.stdout, run.stdout[-3000:] + run.stderr[-3000:]
    assert run.stdout.count(" ok ") >= 6, run.stdout


def test_native_orchestrator_reduce_scatter_on_one_device():
    """cbet_ray_tracing with CBET_FORCE_RCCL=1: cached one-device communicator, ncclReduceScatter into the slab, slab
    download and host +=, twice (the second call reuses the communicator); equal to the plain one-device result.
    Run in a child process: the variable is read by the library at call time, the RCCL state stays out of pytest."""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import load_inputs, parity_err
from cbet_raytracing_3d_amd import api
bn, r, ne, te = load_inputs()
p = api.default_params(47, nbeams=6)          # 49 haloed planes
shape = (49, 49, 49)
plain = np.zeros(shape); api.ray_tracing(te, r, ne, plain, p, beam_norm=bn[:6], gpus=[0])
os.environ["CBET_FORCE_RCCL"] = "1"
for k in range(2):
    forced = np.zeros(shape); api.ray_tracing(te, r, ne, forced, p, beam_norm=bn[:6], gpus=[0])
    err = parity_err(forced, plain)
    print("call %%d: max rel err %%.2e" %% (k, err))
    assert err < 1e-11, err
print("NATIVE RCCL OK")
''' % (ROOT, ROOT)
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert run.returncode == 0 and "NATIVE RCCL OK" in run.stdout, run.stdout[-2000:] + run.stderr[-3000:]
