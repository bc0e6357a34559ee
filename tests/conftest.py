import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "cbet_raytracing_3d_amd", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_inputs():
    """(beam_norm[60,3], r[443], ne[443], te[443]) exactly as main.cu:249-260 reads them."""
    bn = np.loadtxt(os.path.join(DATA, "omega60_beam_norm.txt"))
    te = np.loadtxt(os.path.join(DATA, "s83177_te.txt"))[:443]
    ne = np.loadtxt(os.path.join(DATA, "s83177_ne.txt"))[:443]
    return bn, ne[:, 0].copy(), ne[:, 1].copy(), te[:, 1].copy()


@pytest.fixture(scope="session")
def inputs():
    return load_inputs()


@pytest.fixture(scope="session")
def oracle():
    from oracle import cbet_oracle
    cbet_oracle.lib()
    return cbet_oracle


def parity_err(a, b):
    """SURVEY.md 8(c) parity metric: max_i |a_i-b_i| / max(|b_i|, 1e-9*max_j|b_j|)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    floor = 1e-9 * np.abs(b).max()
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


NCPU = max(1, min(16, os.cpu_count() or 1))
