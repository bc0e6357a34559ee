"""One rank's share of a CBET iteration of the slab-owned loop (tracer.cbet_fixed_point_slabs), timed on ONE GPU.

W ranks (default 8), 256^3, 60 beams.  A steady-state iteration of rank r is
    (1) the energy-field pass of its own beams with the current gain,
    (2) exchange 1: its beams' energy field over every other rank's x-slab out, every other rank's beams over its slab in,
    (3) the gain update of all 60 beams on its own x-slab,
    (4) exchange 2: the new gain of every other rank's beams over its slab out, its own beams' gain over the other slabs in.
(1) and (3) are measured here with the real kernels on real fields (a whole single-GPU first pass and gain update set
the stage; the rank's slab arrays are then cut out of them); (2) and (4) cannot run on one GPU and are PRICED: the bytes
each exchange moves per peer -- dense, and sparse: the 64-byte z-runs the rank's beams can ever touch (tracer.SegmentPlan,
from the beams' bookkeeping-mode footprint), which is what the loop sends -- over one xGMI link per peer at a stated rate,
all seven links busy at once; the pack / unpack kernels of the sparse exchange (cbet_pack_segments / cbet_unpack_segments)
and the strided copies of the dense one are timed locally.  The single-GPU iteration beside it: the energy-field pass of
all 60 beams + the gain update of the whole grid.

usage: python scripts/cbet_rank_share.py [W=8] [n=256] [link GB/s per direction = 64]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cbet_raytracing_3d_amd import api  # noqa: E402
from cbet_raytracing_3d_amd.tracer import RayTracer, _frozen, _parts  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
LINK = float(sys.argv[3]) if len(sys.argv) > 3 else 64.0     # GB/s per direction per peer link, what RCCL send/recv is assumed to sustain
nb = 60
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(n, nbeams=nb), r, ne, te)
gp = api.default_gain_params()
X, Y, Z = tr.grid_shape
plane = Y * Z
dev = tr.device


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


# ---- the stage: a whole single-GPU first pass (directions) and gain update -------------------------------------------
fields, gain = tr.new_fields(), tr.new_grid(per_beam=True)
scratch = torch.empty_like(gain)
change = torch.zeros(2, dtype=torch.float64, device=dev)
tr.tabulate()
tr.launch_cbet(fields, gp, fields=True)
tr.gain_field(fields, gain, gp, change, scratch=scratch)
torch.cuda.synchronize()

# the single-GPU iteration (what one rank's share is compared with)
def energy_pass_all():
    fields[0].zero_()
    tr.launch_cbet(fields[0], gp, fields="energy", gain=gain)


t_field_1 = timed(energy_pass_all)
energy = fields[0].clone()                      # the raw energy field of every beam (the gain kernel normalises in place)
def gain_all():
    fields[0].copy_(energy)
    tr.gain_field(fields, gain, gp, change, scratch=scratch, frozen=True)


t_copy = timed(lambda: fields[0].copy_(energy))
t_gain_1 = timed(gain_all) - t_copy
print("grid %d^3, %d beams, %d ranks; link rate assumed %.0f GB/s per direction per peer" % (n, nb, W, LINK))
print("single GPU, one iteration: energy-field pass %.2f ms + gain update %.2f ms = %.2f ms" % (t_field_1, t_gain_1, t_field_1 + t_gain_1))

beams, slabs = _parts(nb, W), _parts(X, W)
# the footprint of every beam (bookkeeping mode: no ray stops before it leaves the grid) -> the sparse exchange's runs
from cbet_raytracing_3d_amd.tracer import _segment_rows  # noqa: E402
d = tr.derived
foot = torch.zeros((nb, X, Y, Z), dtype=torch.float64, device=dev)
api.trace_nodes(0, d.nindices, None, None, foot, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, d.xconst, d.yconst,
                d.zconst, tr.params.copy(absorption=0, per_beam_grids=1, beam_lo=0, beam_hi=nb), tr.ctx,
                torch.cuda.current_stream(dev).cuda_stream)
support = foot != 0
del foot
torch.cuda.empty_cache()
zs = (Z + 7) // 8
print("footprint of a beam: %.1f %% of the grid's nodes on average; as 64-byte z-runs %.1f %% of the dense volume"
      % (100.0 * support.float().mean().item(),
         100.0 * 8 * sum(_segment_rows(support[b:b + 1], 0, X).shape[0] for b in range(0, nb, 7)) / (len(range(0, nb, 7)) * X * Y * Z)))
rows = []
for rank in sorted({0, W // 2 - 1 if W > 1 else 0, W - 1}):
    (b0, b1), (x0, x1) = beams[rank], slabs[rank]
    nbr, xr = b1 - b0, x1 - x0
    own_e = torch.zeros((nbr, X, Y, Z), dtype=torch.float64, device=dev)
    gain_own = gain[b0:b1].contiguous()

    def field_share():
        own_e.zero_()
        tr.launch_cbet(own_e, gp, fields="energy", gain=gain_own, beam_lo=b0, beam_hi=b1, grid_beam0=b0, grid_beams=nbr)

    t_field = timed(field_share)
    slab_fields = fields[:, :, x0:x1].contiguous()
    slab_fields[0].copy_(energy[:, x0:x1])
    slab_e = slab_fields[0].clone()
    gain_slab = gain[:, x0:x1].contiguous()
    scratch_slab = torch.empty_like(gain_slab)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def gain_share():
        slab_fields[0].copy_(slab_e)
        api.gain_field_packed(slab_fields, None, gain_slab, scratch_slab, change, x0, x1, tr.params, _frozen(gp, True), tr.ctx, stream)

    t_c = timed(lambda: slab_fields[0].copy_(slab_e))
    t_gain = timed(gain_share) - t_c
    # ---- the exchanges: what moves, per peer ----------------------------------------------------------------------
    peers = [s for s in range(W) if s != rank]
    dense_out = [8.0 * nbr * (slabs[s][1] - slabs[s][0]) * plane for s in peers]          # my beams over slab s
    dense_in = [8.0 * (beams[q][1] - beams[q][0]) * xr * plane for q in peers]            # beams of q over my slab
    out_lists, in_lists = [], []
    for s in peers:
        r_ = _segment_rows(support[b0:b1], *slabs[s]).long()
        out_lists.append(torch.stack([r_[:, 0], ((r_[:, 1] + slabs[s][0]) * Y + r_[:, 2]) * zs + r_[:, 3]], 1).to(torch.int32).contiguous())
        q0, q1 = beams[s]
        r_ = _segment_rows(support[q0:q1], x0, x1).long()
        in_lists.append(torch.stack([r_[:, 0] + q0, (r_[:, 1] * Y + r_[:, 2]) * zs + r_[:, 3]], 1).to(torch.int32).contiguous())
    sparse_out = [64.0 * t.shape[0] for t in out_lists]
    sparse_in = [64.0 * t.shape[0] for t in in_lists]
    stage = torch.empty(8 * max(max(t.shape[0] for t in out_lists), max(t.shape[0] for t in in_lists)), dtype=torch.float64, device=dev)

    def sparse_exchange_local(first):      # the pack and unpack launches of ONE exchange (no transport)
        for o, i_ in zip(out_lists, in_lists):
            if first:      # my beams' energy out of own_e, the peers' beams' energy into my slab
                api.pack_segments(own_e, X * plane, Y, Z, o, o.shape[0], stage, stream)
                api.unpack_segments(slab_fields[0], xr * plane, Y, Z, i_, i_.shape[0], stage, stream)
            else:          # the peers' beams' gain out of my slab, my beams' gain into gain_own
                api.pack_segments(gain_slab, xr * plane, Y, Z, i_, i_.shape[0], stage, stream)
                api.unpack_segments(gain_own, X * plane, Y, Z, o, o.shape[0], stage, stream)

    t_pk1, t_pk2 = timed(lambda: sparse_exchange_local(True)), timed(lambda: sparse_exchange_local(False))
    s0 = peers[0]
    view = own_e[:, slabs[s0][0]:slabs[s0][1]]
    buf = torch.empty(view.shape, dtype=torch.float64, device=dev)
    t_dense_copy = (timed(lambda: buf.copy_(view)) + timed(lambda: view.copy_(buf))) * (W - 1)
    ms = lambda b: b / (LINK * 1e9) * 1e3     # every peer link carries its message at once, each way: the largest message decides
    x_dense = [max(max(dense_out), max(dense_in)), max(max(dense_in), max(dense_out))]
    x_sparse = [max(max(sparse_out), max(sparse_in)), max(max(sparse_in), max(sparse_out))]
    share_dense = t_field + t_gain + ms(x_dense[0]) + ms(x_dense[1]) + 2 * t_dense_copy
    share_sparse = t_field + t_gain + ms(x_sparse[0]) + ms(x_sparse[1]) + t_pk1 + t_pk2
    rows.append((rank, share_dense, share_sparse))
    print("rank %d: beams [%d,%d) planes [%d,%d): energy-field pass %.2f ms, slab gain update %.2f ms" % (rank, b0, b1, x0, x1, t_field, t_gain))
    print("        one exchange, dense : %4.0f MB out, largest message %5.1f MB -> %.2f ms on the links + %.2f ms of strided copies"
          % (sum(dense_out) / 1e6, x_dense[0] / 1e6, ms(x_dense[0]), t_dense_copy))
    print("        one exchange, sparse: %4.0f MB out (%4.1f %% of dense), largest message %5.1f MB -> %.2f ms on the links; pack + unpack kernels %.2f ms (fields), %.2f ms (gain)"
          % (sum(sparse_out) / 1e6, 100.0 * sum(sparse_out) / sum(dense_out), x_sparse[0] / 1e6, ms(x_sparse[0]), t_pk1, t_pk2))
    print("        iteration share: %.2f ms dense (%.2fx of the single-GPU iteration), %.2f ms sparse (%.2fx)"
          % (share_dense, (t_field_1 + t_gain_1) / share_dense, share_sparse, (t_field_1 + t_gain_1) / share_sparse))
    del own_e, slab_fields, slab_e, gain_slab, scratch_slab, buf, stage, out_lists, in_lists
    torch.cuda.empty_cache()
worst_d, worst_s = max(r_[1] for r_ in rows), max(r_[2] for r_ in rows)
print("slowest rank: %.2f ms dense -> %.2fx; %.2f ms sparse -> %.2fx  (target: 1/6 of %.2f ms = %.2f ms)"
      % (worst_d, (t_field_1 + t_gain_1) / worst_d, worst_s, (t_field_1 + t_gain_1) / worst_s, t_field_1 + t_gain_1, (t_field_1 + t_gain_1) / 6))
