"""Multi-rank CBET loops with the real device engine: W ranks share the one GPU (gloo carries the exchanges; RCCL
needs one device per rank), every rank runs RayTracer.cbet_solve in the all-reduce form and in the slab-owned
form, and the combined result must equal a single-rank solve (tests/helpers/cbet_slab_rehearsal.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_loops_equal_single_rank(world):
    port = 29800 + (os.getpid() % 150) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "helpers", "cbet_slab_rehearsal.py"), "32"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    lines = [l for l in run.stdout.splitlines() if "world" in l or "REHEARSAL" in l]
    assert run.returncode == 0 and "REHEARSAL PASS" in run.stdout, "\n".join(lines) + run.stderr[-1500:]
    assert len([l for l in lines if l.rstrip().endswith("ok")]) == 2
