"""Does the ORDER in which a launch walks the beams matter?  (The bundle list is beam-major; ~2 beams are resident at a
time; VERDICT r2 item 8: a beam and its antipode cross the same cells and could share the lines of the record table.)
The caller's beam table is permuted -- the library is untouched -- and the 256^3 pass timed: the table's own order,
antipodal pairs back to back, a nearest-neighbour chain (every beam next to the closest not yet visited), random.
usage: python scripts/beam_order.py [n=256]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
r, ne, te = api.load_s83177()
bn = api.omega60_beam_norm()
d = bn @ bn.T
orders = {"table order": list(range(60))}
seen, pairs = set(), []
for i in range(60):
    if i not in seen:
        j = int(d[i].argmin()); pairs += [i, j]; seen |= {i, j}
orders["antipodal pairs"] = pairs
chain, left = [0], set(range(1, 60))
while left:
    j = max(left, key=lambda k: d[chain[-1], k]); chain.append(j); left.remove(j)
orders["nearest-neighbour chain"] = chain
orders["random"] = list(np.random.default_rng(1).permutation(60))
# round 4: by the axis a beam travels along (x-, then y-, then z-dominant; and the three classes dealt round robin): do the
# record table's and the grid's rows of concurrently traced beams share memory channels more in one order than in another?
dom = np.abs(bn).argmax(1)
by_axis = [i for a_ in range(3) for i in range(60) if dom[i] == a_]
orders["grouped by dominant axis"] = by_axis
cls = [[i for i in range(60) if dom[i] == a_] for a_ in range(3)]
mixed = []
while any(cls):
    for c_ in cls:
        if c_:
            mixed.append(c_.pop(0))
orders["dominant axes dealt round robin"] = mixed
orders["table order (again)"] = list(range(60))
# (grids with padded rows: the time no longer depends on where the grid landed, DESIGN.md 4.4 Placement -- round 3's runs of
# this script could not tell orders apart below the 0.5 ms that placement alone made)
for name, perm in orders.items():
    tr = RayTracer(api.default_params(n), r, ne, te, beam_norm=bn[perm])
    e = tr.new_grid(zpitch=True)
    ts = []
    for rep in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e.zero_(); a.record(); tr.launch(e); b.record(); torch.cuda.synchronize()
        if rep > 1: ts.append(a.elapsed_time(b))
    print("%-24s %.3f ms per pass (min %.3f), sum edep %.10e" % (name, sum(ts) / len(ts), min(ts), float(e.sum())), flush=True)
    tr.close()
