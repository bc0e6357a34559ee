"""What would bundles of rays that END TOGETHER buy?  (After the rim packing the idle lane-steps sit where rays stop reaching
the core: bundles there mix 200-step and 450-step rays.)  Ray lengths come from the CPU oracle (one beam; the plasma is
spherical); inside every 2x2 block of whole 8x8 patches whose lanes are less than `thr` busy the 256 rays are sorted by
length and cut into four bundles (a ray keeps its patch lane where free), the list is handed to the context
(cbet_context_set_launch_list) and the 256^3 pass timed against the shipped list.
usage: python scripts/regroup_by_length.py [thr=0.97] [halves]"""
import os, sys
from multiprocessing import Pool
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_inputs
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
from oracle import cbet_oracle as O
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.97
n = 256
bn, r, ne, te = load_inputs()
cfg = O.default_config(n)
p = api.default_params(n)
d = api.derive(p)
live = api.live_ray_list(p).reshape(-1, 64)


def lengths(i):
    return [len(O.ray_path(cfg, bn, r, ne, te, 0, int(q))) if q >= 0 else 0 for q in live[i]]


def timed(tr, e, reps=8):
    ts = []
    for k in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e.zero_(); a.record(); tr.launch(e); b.record(); torch.cuda.synchronize()
        if k > 1: ts.append(a.elapsed_time(b))
    return sum(ts) / len(ts), min(ts)


if __name__ == "__main__":
    with Pool(min(16, os.cpu_count() or 1)) as pool:
        L = np.array(pool.map(lengths, range(len(live)), chunksize=8))
    rpz, zx = p.rays_per_zone, d.zones_spanned
    tile, rem = live // (rpz * rpz), live % (rpz * rpz)
    rx, ry = (tile % zx) * rpz + rem % rpz, (tile // zx) * rpz + rem // rpz
    full = (live >= 0).all(1)
    whole_patch = full & (np.ptp(rx, 1) == 7) & (np.ptp(ry, 1) == 7)     # not a packed rim bundle
    blocks = {}
    for i in np.nonzero(whole_patch)[0]:
        blocks.setdefault((int(rx[i].min()) // 16, int(ry[i].min()) // 16), []).append(int(i))
    util = lambda rows: L[rows].sum() / (64.0 * L[rows].max(1).sum())
    new = live.copy()
    done = 0
    halves = len(sys.argv) > 2 and sys.argv[2] == "halves"
    if halves:
        # gentler: two tangentially adjacent patches exchange halves -- the 32 longest rays of both in one bundle, the 32
        # shortest in the other (roughly their outer and inner halves: a footprint of 8 x 16 rays, contiguous)
        by_pos = {(int(rx[i].min()) // 8, int(ry[i].min()) // 8): int(i) for i in np.nonzero(whole_patch)[0]}
        used = set()
        for (px, py), i in sorted(by_pos.items()):
            if i in used or util([i]) >= thr:
                continue
            cx, cy = 8 * px + 4 - d.nrays_x / 2, 8 * py + 4 - d.nrays_y / 2
            cand = [(px, py + 1), (px, py - 1)] if abs(cx) > abs(cy) else [(px + 1, py), (px - 1, py)]
            j = next((by_pos[c] for c in cand if c in by_pos and by_pos[c] not in used and util([by_pos[c]]) < thr), None)
            if j is None:
                continue
            used |= {i, j}
            rows = [i, j]
            ids, lens, xs, ys = live[rows].ravel(), L[rows].ravel(), rx[rows].ravel(), ry[rows].ravel()
            oi, oj = np.argsort(-lens[:64], kind="stable"), 64 + np.argsort(-lens[64:], kind="stable")
            for row, sel in ((i, np.concatenate([oi[:32], oj[:32]])), (j, np.concatenate([oi[32:], oj[32:]]))):
                bundle, extra = -np.ones(64, dtype=np.int64), []
                for k in sel:
                    lane = (xs[k] & 7) + 8 * (ys[k] & 7)
                    if bundle[lane] < 0: bundle[lane] = ids[k]
                    else: extra.append(ids[k])
                free = np.nonzero(bundle < 0)[0]
                bundle[free[:len(extra)]] = extra
                new[row] = bundle
            done += 2
        blocks = {}
    for key, rows in blocks.items():
        if len(rows) < 2 or util(rows) >= thr:
            continue
        ids, lens, xs, ys = live[rows].ravel(), L[rows].ravel(), rx[rows].ravel(), ry[rows].ravel()
        o = np.argsort(-lens, kind="stable")
        for g, row in enumerate(rows):
            sel = o[64 * g:64 * g + 64]
            bundle, extra = -np.ones(64, dtype=np.int64), []
            for k in sel:
                lane = (xs[k] & 7) + 8 * (ys[k] & 7)
                if bundle[lane] < 0: bundle[lane] = ids[k]
                else: extra.append(ids[k])
            free = np.nonzero(bundle < 0)[0]
            bundle[free[:len(extra)]] = extra
            new[row] = bundle
        done += len(rows)
    steps = {int(i): int(s) for i, s in zip(live.ravel(), L.ravel()) if i >= 0}
    Ln = np.array([[steps[int(i)] if i >= 0 else 0 for i in b] for b in new])
    print("threshold %.2f: %d of %d bundles regrouped; lane utilisation %.4f -> %.4f (wave-steps per beam %d -> %d)"
          % (thr, done, len(live), L.sum() / (64.0 * L.max(1).sum()), Ln.sum() / (64.0 * Ln.max(1).sum()), L.max(1).sum(), Ln.max(1).sum()), flush=True)
    tr = RayTracer(p, r, ne, te)
    e = tr.new_grid()
    t0 = timed(tr, e); ref = e.clone(); c0 = tr.counters(reset=True)
    tr.ctx.set_launch_list(new.ravel())
    t1 = timed(tr, e); c1 = tr.counters(reset=True)
    tr.ctx.set_launch_list(live.ravel())
    t2 = timed(tr, e)
    print("shipped list %.3f ms (min %.3f) | regrouped %.3f ms (min %.3f) | shipped again %.3f ms; max rel diff of the grids %.2e; ray-steps %d / %d"
          % (t0[0], t0[1], t1[0], t1[1], t2[0], float(((e - ref).abs() / ref.abs().max()).max()), c0.ray_steps, c1.ray_steps))
