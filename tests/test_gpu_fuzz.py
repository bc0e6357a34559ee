"""Randomised parity sweep (tests/helpers/fuzz_parity.py): small random grids, beam subsets, rays per zone, absorption,
sharding, beam-resolved grids, all three kernel formulations -- each case against the CPU oracle."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_match_the_oracle(seed):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "fuzz_parity.py"), "30", str(seed)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    tail = "\n".join(run.stdout.splitlines()[-12:])
    assert run.returncode == 0, tail + run.stderr[-2000:]
    assert "failures 0" in tail
