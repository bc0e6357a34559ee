"""CPU study (oracle ray paths): how wide, in cells, does an 8x8-ray bundle get along its path?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_inputs
from cbet_raytracing_3d_amd import api
from oracle import cbet_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bn, r, ne, te = load_inputs()
cfg = O.default_config(n); p = api.default_params(n)
slots = api.live_ray_list(p).reshape(-1, 64)
rng = np.random.default_rng(1)
hist = np.zeros(40, dtype=np.int64); tot = 0; missed = 0; lanes = 0
phase = np.zeros((2, 12))
for b in rng.choice(len(slots), 60, replace=False):
    beam = int(rng.integers(60))
    ids = slots[b][slots[b] >= 0]
    paths = [O.ray_path(cfg, bn, r, ne, te, beam, int(i)) for i in ids]
    L = max(len(q) for q in paths)
    for t in range(L):
        cells = np.array([q[t, 3:6] for q in paths if len(q) > t])
        span = (cells.max(0) - cells.min(0)).max()
        hist[int(span)] += 1; tot += 1
        # lanes further than 3 cells (any axis) from the per-axis median: cannot fit a 7-wide box centred on the bulk
        med = np.median(cells, axis=0)
        far = (np.abs(cells - med) > 3).any(1).sum()
        missed += far; lanes += len(cells)
        ph = min(11, int(12 * t / L)); phase[0, ph] += far; phase[1, ph] += len(cells)
print("n=%d  wave-steps %d" % (n, tot))
print("max-axis cell span histogram (span: share):", {i: round(h / tot, 3) for i, h in enumerate(hist) if h})
print("share of wave-steps with span > 5: %.3f   > 7: %.3f" % (hist[6:].sum() / tot, hist[8:].sum() / tot))
print("lanes > 3 cells from the bundle median: %.4f of lane-steps" % (missed / lanes))
print("by path phase (12 bins):", np.round(phase[0] / np.maximum(1, phase[1]), 3))
