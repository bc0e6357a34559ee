"""One rank's share of a CBET iteration of the slab-owned loop, run AS THE PRODUCT ISSUES IT on one GPU.

tracer.cbet_fixed_point_slabs itself runs here for rank r of W (default 8; 256^3, 60 beams): its beam groups on the two
alternating trace streams, its per-beam grouped send/recvs on the communication stream behind the trace events, the slab
gain update behind the arrival fence, the gain coming back group by group while the next pass already traces, the
convergence scalars through the host.  The one thing one GPU cannot do is the transport; tracer._SlabExchanger takes a
stand-in for it (`emulate`): on the communication stream, exactly where RCCL's grouped send/recv would run, it
  * supplies what the peers would send -- their beams' fields over this rank's slab and this rank's beams' gain over their
    slabs, cut out of a converged single-GPU solve -- with device copies into the very views the receives target, and
  * holds the stream for the PRICED duration of the call: every message of a grouped call travels on its own peer link,
    all links at once, each way: max over peers of max(bytes out, bytes in) / link rate + a fixed cost per call.
Kernels, copies, stream order and host round trips are real; only the link time is a model, and it is stated.
The slabs are cut by gain-update work from the reference solve's beam counts, as the loop cuts them.

usage: python scripts/cbet_rank_share.py [W=8] [n=256] [link GB/s per direction per peer = 64] [trace groups = 4] [ranks = 0,3,7] [slab layout = equal | paired | halves | a number > 1] [channels = 1 | 2]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api  # noqa: E402
from cbet_raytracing_3d_amd import tracer as T  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
LINK = float(sys.argv[3]) if len(sys.argv) > 3 else 64.0      # GB/s per direction per peer link, what RCCL send/recv is assumed to sustain
GROUPS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
RANKS = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else sorted({0, W // 2 - 1 if W > 1 else 0, W - 1})
CHANNELS = int(sys.argv[7]) if len(sys.argv) > 7 else 1        # 2: exchange 2 on a stream of its own (tracer.cbet_fixed_point_slabs two_channels)
LAYOUT = sys.argv[6] if len(sys.argv) > 6 else "equal"        # tracer.slab_pieces: "paired", "equal", or the widest slab of a work-balanced cut (x the equal share)
LAYOUT = LAYOUT if LAYOUT in ("paired", "equal", "halves") else float(LAYOUT)
CALL_US = 15.0                                                 # fixed cost of one grouped send/recv (launch + handshake)
nb = 60
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 200))
dist.init_process_group("gloo", rank=0, world_size=1)          # the loop's scalar all-reduce / broadcast: trivial, through the host
r, ne, te = api.load_s83177()
tr = T.RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
gp = api.default_gain_params()
X, Y, Z = tr.grid_shape
plane = Y * Z
dev = tr.device

# cycles of torch.cuda._sleep per millisecond
def _sleep_rate():
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    a.record(); torch.cuda._sleep(20_000_000); b.record(); torch.cuda.synchronize()
    return 20_000_000 / a.elapsed_time(b)
SLEEP_PER_MS = _sleep_rate()


def timed(fn, reps=3):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


# ---- the reference: a converged single-GPU solve; its raw energy fields and gain are what the peers "send" -----------
edep = tr.new_grid()
ws = torch.empty(api.cbet_workspace_bytes(tr.params) // 8, dtype=torch.float64, device=dev)
rep0 = api.cbet_solve(tr.d_te, tr.d_r, tr.d_ne, edep, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, tr.params, gp,
                      workspace=ws, ctx=tr.ctx, stream=torch.cuda.current_stream().cuda_stream)
hs = X * plane
fields = ws[: 4 * nb * hs].view((4, nb) + tr.grid_shape)
gain = ws[4 * nb * hs: 5 * nb * hs].view((nb,) + tr.grid_shape)
change = torch.zeros(2, dtype=torch.float64, device=dev)


def energy_pass_all():
    fields[0].zero_()
    tr.launch_cbet(fields[0], gp, fields="energy", gain=gain)


t_field_1 = timed(energy_pass_all)
energy = fields[0].clone()                      # raw energy field of every beam with the converged gain
t_copy = timed(lambda: fields[0].copy_(energy))
t_gain_1 = timed(lambda: (fields[0].copy_(energy), tr.gain_field(fields, gain, gp, change, pair_once=True, frozen=True))) - t_copy
ref_gain = gain.clone()
single = t_field_1 + t_gain_1
print("grid %d^3, %d beams, %d ranks, %d trace groups per pass; links priced at %.0f GB/s per direction per peer + %.0f us per grouped call"
      % (n, nb, W, GROUPS, LINK, CALL_US))
print("single GPU, one iteration: energy-field pass %.2f ms + gain update %.2f ms = %.2f ms  (solve: %d passes)"
      % (t_field_1, t_gain_1, single, rep0.passes))
counts = (energy != 0).sum(0).to(torch.int32)
weights = T.gain_update_weights(counts)
beams = T._parts(nb, W)
spread = lambda pcs: max(sum(weights[a:b].sum() for a, b in p_) for p_ in pcs) / (weights.sum() / W)
show = lambda pcs: " ".join("+".join("%d" % (b - a) for a, b in p_) for p_ in pcs)
for name, lay in (("one equal slab per rank", "equal"), ("paired blocks (edge + centre)", "paired"), ("one slab by gain-update work", 9.0), ("... none wider than 1.15 x", 1.15)):
    pcs = T.slab_pieces(lay, X, W, weights)
    print("%-32s planes %s  (modelled gain-update cost, max/mean: %.2f)%s" % (name, show(pcs), spread(pcs), "   <- this run" if lay == LAYOUT else ""))
del counts


class EmulatedEngine(T._DeviceCbetEngine):
    """The device engine of rank `rank`, with the peers' side of the world supplied from the reference solve."""

    def __init__(self, rank, link):
        super().__init__(tr, tr.new_grid(), gp)
        self.rank, self.link = rank, link
        self.emulate_transport = self.transport
        self.link_ms = self.calls = self.link_steady = 0.0
        self.passes_steady = 0
        self.marks = []
        self.trace_ends = {}
        self.host_t = []

    def presence_counts(self):           # what the all-reduce over the ranks would return
        return (energy != 0).sum(0).to(torch.int32)

    def begin_beams(self, b0, b1):
        super().begin_beams(b0, b1)
        self.gain_own.copy_(ref_gain[b0:b1])        # start from the converged gain: every pass is a steady-state pass

    def transport(self, xch, sends, recvs):
        # the arrivals: device copies on a side stream BESIDE the priced link time (a real receive writes while it travels)
        comm = torch.cuda.current_stream()
        if not hasattr(self, "aux"):
            self.aux = torch.cuda.Stream()
        start = torch.cuda.Event()
        start.record(comm)
        self.aux.wait_event(start)
        out, inn = {}, {}
        for t, peer in sends:
            out[peer] = out.get(peer, 0) + 8 * t.numel()
        with torch.cuda.stream(self.aux):
            mine = {sf.untyped_storage().data_ptr(): k for k, sf in enumerate(self.slab_fields)}
            for t, peer in recvs:
                inn[peer] = inn.get(peer, 0) + 8 * t.numel()
                off = t.storage_offset()
                k = mine.get(t.untyped_storage().data_ptr())
                if k is not None:                                                                         # a peer's beam over my piece k
                    lo, hi = self.pieces[k]
                    c, b = off // (nb * (hi - lo) * plane), (off // ((hi - lo) * plane)) % nb
                    t.copy_((energy if c == 0 else fields[c])[b, lo:hi])
                else:                                                                                     # my beam's gain over a peer's piece
                    i, xs0 = off // (X * plane), (off % (X * plane)) // plane
                    t.copy_(ref_gain[self.b0 + i, xs0:xs0 + t.shape[0]])
            arrived = torch.cuda.Event()
            arrived.record(self.aux)
        ms = (max(list(out.values()) + list(inn.values())) / (self.link * 1e9) * 1e3 if self.link > 0 else 0.0) + CALL_US * 1e-3
        torch.cuda._sleep(int(ms * SLEEP_PER_MS))
        comm.wait_event(arrived)
        self.link_ms += ms
        if len(self.marks) >= 2:              # steady state: the energy field alone moves
            self.link_steady += ms
        self.calls += 1

    def trace_group(self, i0, i1, use_gain, full=True, wait=()):
        if not self.trace_ends.get(len(self.marks)):
            self.host_t.append(["start", time.perf_counter()])
        done = super().trace_group(i0, i1, use_gain, full, wait)
        st = self.s_trace[(self._launches - 1) % len(self.s_trace)]
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(st)                      # right behind the launch on its stream
        self.trace_ends.setdefault(len(self.marks), []).append(ev)
        return done

    def update_gain_slab(self, frozen=False, after_piece=None):
        self.host_t.append(["update", time.perf_counter()])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        out = super().update_gain_slab(frozen, after_piece)
        ev[1].record()
        self.marks.append(ev)
        if len(self.marks) > 2:
            self.passes_steady += 1
        return out


def run_rank(rank, link, groups, layout, passes=7):
    eng = EmulatedEngine(rank, link)
    g = type(gp).from_buffer_copy(gp)
    g.tolerance, g.max_passes = 0.0, passes
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = T.cbet_fixed_point_slabs(eng, g, nb, X, rank, W, None, trace_groups=groups, slab_layout=layout, two_channels=CHANNELS == 2)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    starts = [m[0] for m in eng.marks]
    per_pass = [starts[k].elapsed_time(starts[k + 1]) for k in range(2, len(starts) - 1)]      # steady state: update k -> update k + 1
    t_gain = sum(m[0].elapsed_time(m[1]) for m in eng.marks[2:]) / len(eng.marks[2:])
    # where a steady-state pass spends its time: update k ends -> last trace of pass k + 1 ends -> update k + 1 starts -> ends
    seg = [[], [], []]
    for k in range(2, len(eng.marks) - 1):
        last = eng.trace_ends[k + 1][-1]
        seg[0].append(eng.marks[k][1].elapsed_time(last))                 # gain back (first group), host round trip, all traces
        seg[1].append(last.elapsed_time(eng.marks[k + 1][0]))             # the tail of exchange 1 that nothing hides
        seg[2].append(eng.marks[k + 1][0].elapsed_time(eng.marks[k + 1][1]))
    seg = [sum(v) / len(v) for v in seg]
    ht = eng.host_t
    enq = [1e3 * (ht[j + 1][1] - ht[j][1]) for j in range(len(ht) - 1) if ht[j][0] == "start" and ht[j + 1][0] == "update"][2:]
    gap = [1e3 * (ht[j + 1][1] - ht[j][1]) for j in range(len(ht) - 1) if ht[j][0] == "update" and ht[j + 1][0] == "start"][2:]
    seg += [sum(enq) / max(1, len(enq)), sum(gap) / max(1, len(gap))]
    out = dict(rank=rank, iteration=sum(per_pass) / len(per_pass), gain=t_gain, slab=rep["slabs"][rank], beams=beams[rank],
               segments=seg, link_ms_per_pass=eng.link_steady / max(1, eng.passes_steady), calls_per_pass=eng.calls / passes, wall=wall, bytes=eng.slab_bytes())
    eng.tr = None
    del eng
    torch.cuda.empty_cache()
    return out


def trace_alone(rank, groups):
    """The energy-field pass of the rank's beams alone: one launch, and in `groups` launches on alternating streams."""
    eng = T._DeviceCbetEngine(tr, tr.new_grid(), gp)
    b0, b1 = beams[rank]
    eng.begin_beams(b0, b1)
    eng.gain_own.copy_(ref_gain[b0:b1])
    idx = [g for g in T._parts(max(q1 - q0 for q0, q1 in beams), groups)]

    def one():
        torch.cuda.current_stream().wait_event(eng.trace_group(0, b1 - b0, True, False))

    def grouped():
        for i0, i1 in idx:
            j0, j1 = min(i0, b1 - b0), min(i1, b1 - b0)
            if j1 > j0:
                torch.cuda.current_stream().wait_event(eng.trace_group(j0, j1, True, False))

    def single():
        j1 = min(idx[0][1], b1 - b0)
        torch.cuda.current_stream().wait_event(eng.trace_group(0, j1, True, False))

    a, b, c = timed(one), timed(grouped), timed(single)
    del eng
    torch.cuda.empty_cache()
    return a, b, c


def model_iteration(G, t_single, t_grouped, call_ms, update_ms, host_ms=0.25, enqueue_ms=0.12):
    """Critical path of one steady-state pass in the order tracer.cbet_fixed_point_slabs enqueues it, from measured pieces:
    one in-order communication channel (exchange 2's G grouped calls... per group, then each group's exchange-1 calls
    behind its trace), traces gated by their gain's arrival and overlapping on the rotating streams (a group alone takes
    t_single, G of them back to back finish every t_grouped / G), the update behind the last arrival, the host resuming
    host_ms after the update and enqueueing one item per enqueue_ms.  call_ms = priced duration of the grouped calls of ONE
    group (all its beams), either exchange.  Returns the period [ms]."""
    t = 0.0                     # the previous update ends
    comm = 0.0
    gain_in = []
    for k in range(G):          # exchange 2, enqueued before the host sync
        comm = max(comm, t) + call_ms
        gain_in.append(comm)
    host = t + host_ms
    prev_end, ends = 0.0, []
    for k in range(G):
        host += enqueue_ms
        start = max(host, gain_in[k])
        end = max(start + t_single, prev_end + t_grouped / G)
        ends.append(end)
        prev_end = end
        host += enqueue_ms
    for k in range(G):          # exchange 1: group k's calls behind its trace, in order
        comm = max(comm, ends[k]) + call_ms
    return comm + update_ms


rows = []
for rank in RANKS:
    t_one, t_grp, t_single = trace_alone(rank, GROUPS)
    run_rank(rank, LINK, GROUPS, LAYOUT, passes=4)     # warm-up: allocations, streams
    res = run_rank(rank, LINK, GROUPS, LAYOUT)
    if os.environ.get("CBET_SHARE_MAIN_ONLY"):
        free = flat = res
    else:
        free = run_rank(rank, 0.0, GROUPS, LAYOUT)     # the same schedule with free links: what the exchanges still cost in it
        flat = run_rank(rank, LINK, 1, "equal")        # one group, one equal slab: nothing overlapped, the round-3 shape
    alts = {} if os.environ.get("CBET_SHARE_MAIN_ONLY") else \
        {(gr, ba): run_rank(rank, LINK, gr, ba)["iteration"] for gr in (4, 8) for ba in ("equal", "paired", "halves", 1.15, 9.0) if (gr, ba) != (GROUPS, LAYOUT)}
    alts[(GROUPS, LAYOUT)] = res["iteration"]
    rows.append(res)
    print("rank %d: beams [%d,%d), planes %s  -- workspace %.2f GB" % (rank, *res["beams"], " + ".join("[%d,%d)" % pc for pc in res["slab"]), res["bytes"] / 1e9))
    print("        energy-field pass of its beams alone: %.2f ms in one launch, %.2f ms in %d groups on rotating streams; slab gain update %.2f ms"
          % (t_one, t_grp, GROUPS, res["gain"]))
    print("        steady-state iteration as issued: %.2f ms  (%.2fx of the single-GPU iteration); priced link time in it %.2f ms over %.0f grouped calls"
          % (res["iteration"], single / res["iteration"], res["link_ms_per_pass"], res["calls_per_pass"]))
    print("        ... of which: update done -> last trace of the next pass done %.2f ms | -> its update starts (exchange 1's exposed tail) %.2f ms | update %.2f ms"
          % tuple(res["segments"][:3]))
    print("        ... host: enqueueing a pass (first trace -> update call) takes %.2f ms, update call -> next pass's first trace (exchange-2 head, scalars through the host) %.2f ms"
          % tuple(res["segments"][3:]))
    nbr_, xr_ = res["beams"][1] - res["beams"][0], max(sum(b_ - a_ for a_, b_ in p_) for p_ in T.slab_pieces(LAYOUT, X, W, weights))
    per_beam_call = 8.0 * xr_ * plane / (LINK * 1e9) * 1e3 + CALL_US * 1e-3
    for gm in sorted({2, 4, 8, GROUPS}):
        call = per_beam_call * -(-max(q1 - q0 for q0, q1 in beams) // gm)
        tg = t_grp if gm == GROUPS else None
        if tg is None:
            continue
        print("        ... critical path of the issued order from the measured pieces (%d groups: one alone %.2f ms, all %.2f ms; %.2f ms of link per group and exchange; update %.2f ms): %.2f ms (%.2fx)"
              % (gm, t_single, tg, call, res["gain"], model_iteration(gm, t_single, tg, call, res["gain"]), single / model_iteration(gm, t_single, tg, call, res["gain"])))
    print("        ... with free links: %.2f ms -> the exchanges cost %.2f ms of the iteration (their copies, calls and what the schedule cannot hide)"
          % (free["iteration"], res["iteration"] - free["iteration"]))
    print("        ... one group, one equal slab per rank (nothing overlapped): %.2f ms (%.2fx), slab gain update %.2f ms"
          % (flat["iteration"], single / flat["iteration"], flat["gain"]))
    print("        ... iteration [ms] by (trace groups, slab layout):", "  ".join("(%d, %s) %.2f" % (k[0], k[1], v) for k, v in sorted(alts.items(), key=str)))
worst = max(rows, key=lambda q: q["iteration"])
print("slowest rank %d: %.2f ms -> %.2fx  (target: 1/6 of %.2f ms = %.2f ms)" % (worst["rank"], worst["iteration"], single / worst["iteration"], single, single / 6))
dist.destroy_process_group()
