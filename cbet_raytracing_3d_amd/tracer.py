"""RayTracer: torch-held device buffers around the C-ABI launch, and the multi-GPU step.

torch is plumbing here -- device memory, streams and torch.distributed (backend "nccl" = RCCL over
xGMI).  All computation happens in libcbet_mi355x.so through cbet_raytracing_3d_amd.api.

The multi-GPU scheme replaces main.cu:166-210 (/root/reference): instead of contiguous blocks of
nbeams/nGPUs beams per device and a host-side sum, every rank traces an interleaved 1/world_size
share of the ray bundles of EVERY beam into its private (nx+2)(ny+2)(nz+2) grid and the grids are
summed with one all-reduce.
"""
import numpy as np
import torch

from . import api


class RayTracer:
    """One device's share of a ray-tracing pass.

    Mirrors the device-side state rayTracing() sets up per GPU (main.cu:133-152): the seven small
    read-only arrays uploaded once, a deposition grid, and the launch constants of main.cu:156-159.
    """

    def __init__(self, params, r_profile, ne_profile, te_profile, beam_norm=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("RayTracer needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        self.gpu = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.params = params.copy()
        self.derived = api.derive(self.params)
        if beam_norm is None:
            beam_norm = api.omega60_beam_norm()[: self.params.nbeams]
        beam_norm = np.ascontiguousarray(beam_norm, dtype=np.float64).reshape(-1, 3)
        if beam_norm.shape[0] != self.params.nbeams:
            raise ValueError("beam_norm has %d rows, params.nbeams=%d" % (beam_norm.shape[0], self.params.nbeams))
        phase_r, pow_r = api.host_power_table()
        bbeam = api.host_beam_trig(beam_norm)

        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

        self.d_beam_norm, self.d_bbeam_norm = up(beam_norm), up(bbeam)
        self.d_pow_r, self.d_phase_r = up(pow_r), up(phase_r)
        self.d_r, self.d_ne, self.d_te = up(r_profile), up(ne_profile), up(te_profile)
        for t in (self.d_r, self.d_ne, self.d_te):
            if t.numel() != self.params.nprofile:
                raise ValueError("profile length != params.nprofile")
        self.ctx = api.Context(self.params, self.gpu)
        self.grid_shape = (self.params.nx + 2, self.params.ny + 2, self.params.nz + 2)

    def new_grid(self, per_beam=False):
        """A zeroed deposition grid; per_beam=True: one grid per beam (cbet_params.per_beam_grids)."""
        shape = ((self.params.nbeams,) + self.grid_shape) if per_beam else self.grid_shape
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def launch(self, edep, shard_index=0, shard_count=1, beam_lo=0, beam_hi=None,
               kernel_variant=None, force_wide_index=None, use_host_trig=True):
        """Enqueue one launch_ray_XYZ on torch's current stream, accumulating into `edep`."""
        per_beam = edep.dim() == 4
        want = ((self.params.nbeams,) + self.grid_shape) if per_beam else self.grid_shape
        if edep.dtype != torch.float64 or not edep.is_contiguous() or tuple(edep.shape) != want:
            raise ValueError("edep must be a contiguous float64 tensor of shape %s (or nbeams x that)" % (self.grid_shape,))
        p = self.params.copy(per_beam_grids=1 if per_beam else 0, beam_lo=beam_lo,
                             beam_hi=self.params.nbeams if beam_hi is None else beam_hi,
                             shard_index=shard_index, shard_count=shard_count)
        if kernel_variant is not None:
            p.kernel_variant = kernel_variant
        if force_wide_index is not None:
            p.force_wide_index = force_wide_index
        d = self.derived
        stream = torch.cuda.current_stream(self.device).cuda_stream
        api.launch_ray_XYZ(0, d.nindices, self.d_te, self.d_r, self.d_ne, edep,
                           self.d_bbeam_norm if use_host_trig else None, self.d_beam_norm,
                           self.d_pow_r, self.d_phase_r, d.xconst, d.yconst, d.zconst, p,
                           ctx=self.ctx, stream=stream)
        return edep

    def counters(self, reset=False):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        return self.ctx.counters(stream, reset)

    # ---- CBET stage (parity unpinned, see include/cbet_mi355x.h) --------------
    def tabulate(self):
        """Fill the context's node tables from the radial profiles (what launch() does first)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        api.tabulate_plasma(self.ctx, self.params, self.d_te, self.d_r, self.d_ne, stream)

    def launch_cbet(self, out, gain_params, fields=False, gain=None, beam_gain=None, shard_index=0,
                    shard_count=1, ne3d=None, kappa3d=None, beam_lo=0, beam_hi=None):
        """One trace with the CBET hooks on torch's current stream (node tables must be filled:
        tabulate(), or pass ne3d / kappa3d).  fields=False: deposit the absorbed energy into `out`
        ((n+2)^3 grid or nbeams of them); fields=True: the fused field pass, `out` = new_fields()."""
        if fields:
            want = (4, self.params.nbeams) + self.grid_shape
            per_beam = True
        else:
            per_beam = out.dim() == 4
            want = ((self.params.nbeams,) + self.grid_shape) if per_beam else self.grid_shape
        if out.dtype != torch.float64 or not out.is_contiguous() or tuple(out.shape) != want:
            raise ValueError("out must be a contiguous float64 tensor of shape %s" % (want,))
        if gain is not None and (gain.dtype != torch.float64 or not gain.is_contiguous() or
                                 tuple(gain.shape) != (self.params.nbeams,) + self.grid_shape):
            raise ValueError("gain must be a contiguous float64 tensor of shape nbeams x %s" % (self.grid_shape,))
        p = self.params.copy(per_beam_grids=1 if per_beam else 0, beam_lo=beam_lo,
                             beam_hi=self.params.nbeams if beam_hi is None else beam_hi,
                             shard_index=shard_index, shard_count=shard_count)
        d = self.derived
        stream = torch.cuda.current_stream(self.device).cuda_stream
        api.trace_cbet(0, d.nindices, ne3d, kappa3d, gain, api.DEPOSIT_FIELDS if fields else api.DEPOSIT_ENERGY, out,
                       beam_gain, self.d_bbeam_norm, self.d_beam_norm, self.d_pow_r, self.d_phase_r, d.xconst,
                       d.yconst, d.zconst, p, gain_params, self.ctx, stream)
        return out

    def new_fields(self):
        """Zeroed [4][nbeams][(n+2)^3] field array (energy x path length, energy x displacement x/y/z)."""
        return torch.zeros((4, self.params.nbeams) + self.grid_shape, dtype=torch.float64, device=self.device)

    def gain_field(self, fields, gain, gain_params, change=None, ne3d=None, scratch=None, x_lo=0, x_hi=None):
        """Normalise `fields` in place and relax `gain` towards the gain coefficient they imply.
        scratch: a work array shaped like `gain` (each beam pair evaluated once), or None (ordered kernel).
        x_lo, x_hi: only the planes [x_lo, x_hi) of the deposit grid (one rank's slab)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        api.gain_field_slab(fields, ne3d, gain, scratch, change, x_lo, self.grid_shape[0] if x_hi is None else x_hi,
                            self.params, gain_params, self.ctx, stream)
        return gain

    def cbet_solve(self, edep, gain_params, rank=0, world_size=1, group=None, fields=None, gain=None, slabs=False):
        """The CBET iteration, one rank's share (cbet_fixed_point -- or, with slabs=True, cbet_fixed_point_slabs,
        the exchange sized for xGMI -- with this device as the engine): the deposition pass is ADDED into
        `edep` (not reduced here: use allreduce_grid).  Single-rank callers can use the native loop instead:
        api.cbet_solve."""
        engine = _DeviceCbetEngine(self, edep, gain_params, fields, gain)
        if slabs:
            rep = cbet_fixed_point_slabs(engine, gain_params, self.params.nbeams, self.grid_shape[0], rank, world_size, group)
        else:
            rep = cbet_fixed_point(engine, gain_params, rank, world_size, group)
        rep["gain"] = engine.gain
        return rep

    def node_tables(self):
        """Copies of the context's node tables (ne3d, kappa3d) as numpy arrays, for tests."""
        n = self.params.nx * self.params.ny * self.params.nz
        a, b = self.ctx.tables()
        out = []
        for addr in (a, b):
            h = np.empty(n)
            api.moveToAndFromGPU(h, addr, 8 * n, self.gpu)
            out.append(h.reshape(self.params.nx, self.params.ny, self.params.nz))
        return out

    def close(self):
        self.ctx.close()


def shard_of_rank(rank, world_size):
    """(shard_index, shard_count) of a rank: bundle g belongs to rank g % world_size."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside [0, world_size)")
    return rank, world_size


def allreduce_grid(edep, group=None):
    """Sum the per-rank deposition grids in place (RCCL all-reduce over xGMI with backend
    "nccl"; gloo on CPU tensors in the tests).  Replaces main.cu:178-210.  No-op without an
    initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(edep, op=dist.ReduceOp.SUM, group=group)
    return edep


class _DeviceCbetEngine:
    """The per-rank compute of the CBET iteration on a RayTracer's device (see cbet_fixed_point)."""

    def __init__(self, tracer, edep, gain_params, fields=None, gain=None):
        self.tr, self.edep, self.gp = tracer, edep, gain_params
        self.fields = tracer.new_fields() if fields is None else fields
        self.gain = tracer.new_grid(per_beam=True) if gain is None else gain
        self.scratch = torch.empty_like(self.gain)
        self.change = torch.zeros(2, dtype=torch.float64, device=tracer.device)
        self.beam_gain = torch.zeros(tracer.params.nbeams, dtype=torch.float64, device=tracer.device)

    def begin(self):
        self.tr.tabulate()
        self.gain.zero_()

    def field_passes(self, use_gain, shard_index, shard_count):
        self.fields.zero_()
        self.tr.launch_cbet(self.fields, self.gp, fields=True, gain=self.gain if use_gain else None,
                            shard_index=shard_index, shard_count=shard_count)
        return self.fields

    def update_gain(self, fields):
        self.change.zero_()
        self.tr.gain_field(fields, self.gain, self.gp, self.change, scratch=self.scratch)
        return self.change

    # the slab-owned variant (cbet_fixed_point_slabs): whole beams per rank, the gain update per x-slab
    def field_passes_beams(self, use_gain, beam_lo, beam_hi):
        self.fields.zero_()
        self.tr.launch_cbet(self.fields, self.gp, fields=True, gain=self.gain if use_gain else None,
                            beam_lo=beam_lo, beam_hi=beam_hi)
        return self.fields

    def update_gain_slab(self, fields, x_lo, x_hi):
        self.change.zero_()
        self.tr.gain_field(fields, self.gain, self.gp, self.change, scratch=self.scratch, x_lo=x_lo, x_hi=x_hi)
        return self.change

    def deposit_beams(self, beam_lo, beam_hi):
        self.beam_gain.zero_()
        self.tr.launch_cbet(self.edep, self.gp, gain=self.gain, beam_gain=self.beam_gain, beam_lo=beam_lo, beam_hi=beam_hi)
        return self.beam_gain

    def deposit(self, shard_index, shard_count):
        self.beam_gain.zero_()
        self.tr.launch_cbet(self.edep, self.gp, gain=self.gain, beam_gain=self.beam_gain,
                            shard_index=shard_index, shard_count=shard_count)
        return self.beam_gain


def cbet_fixed_point(engine, gain_params, rank=0, world_size=1, group=None):
    """The CBET fixed-point iteration over `world_size` ranks (parity unpinned).

    Every pass: each rank deposits the four field components of ITS share of the ray bundles (the plain
    pass's interleaved sharding), the fields are summed over ranks with one all-reduce, and every rank
    updates the full gain coefficient from them (redundantly -- it needs all of it for its own rays).
    Stops when sum |dK| / sum |K| < tolerance, then runs the deposition pass and all-reduces the
    per-beam energy balance.  `engine` supplies the per-rank compute:
        begin(); field_passes(use_gain, shard_index, shard_count) -> fields tensor;
        update_gain(fields) -> tensor {sum |dK|, sum |K|}; deposit(shard_index, shard_count) -> beam_gain tensor
    (the device engine is RayTracer.cbet_solve's; the CPU tests drive this loop with an oracle engine).
    Returns {passes, converged, change, beam_gain, imbalance}."""
    si, sc = shard_of_rank(rank, world_size)
    engine.begin()
    rep = {"passes": 0, "converged": False, "change": float("inf")}
    for it in range(gain_params.max_passes):
        fields = engine.field_passes(it > 0, si, sc)
        if world_size > 1:
            allreduce_grid(fields, group)
        ch = engine.update_gain(fields).cpu()
        rep["passes"] = it + 1
        rep["change"] = float(ch[0] / ch[1]) if float(ch[1]) > 0 else 0.0
        if rep["change"] < gain_params.tolerance:
            rep["converged"] = True
            break
    beam_gain = engine.deposit(si, sc)
    if world_size > 1:
        allreduce_grid(beam_gain, group)
    bg = beam_gain.cpu().numpy().copy()
    rep["beam_gain"] = bg
    rep["imbalance"] = float(abs(bg.sum()) / np.abs(bg).sum()) if np.abs(bg).sum() > 0 else 0.0
    return rep


def _parts(total, world_size):
    """Contiguous near-equal parts of range(total): [(lo, hi)] per rank."""
    return [((r * total) // world_size, ((r + 1) * total) // world_size) for r in range(world_size)]


def _exchange(tensor, send_index, recv_index, rank, world_size, group):
    """Point-to-point exchange over xGMI / RCCL (gloo in the CPU tests): rank r sends tensor[send_index(s)] to
    every other rank s and stores what s sends it in tensor[recv_index(s)].  Index tuples select strided views;
    the copies to and from contiguous staging buffers are the pack / unpack of an all-to-all."""
    import torch.distributed as dist
    ops, inbox = [], []
    for s in range(world_size):
        if s == rank:
            continue
        out = tensor[send_index(s)].contiguous()
        buf = torch.empty_like(tensor[recv_index(s)], memory_format=torch.contiguous_format)
        inbox.append((s, buf))
        ops.append(dist.P2POp(dist.isend, out, s if group is None else dist.get_global_rank(group, s), group))
        ops.append(dist.P2POp(dist.irecv, buf, s if group is None else dist.get_global_rank(group, s), group))
    if ops:
        if tensor.is_cuda:   # staging copies done before a backend that is not stream-ordered (gloo) reads them
            torch.cuda.current_stream(tensor.device).synchronize()
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)
    for s, buf in inbox:
        tensor[recv_index(s)] = buf


def cbet_fixed_point_slabs(engine, gain_params, nbeams, nx_halo, rank=0, world_size=1, group=None):
    """The CBET fixed-point iteration with the exchange sized for point-to-point xGMI (parity
    unpinned; same passes and same result as cbet_fixed_point).

    Rank r traces WHOLE beams [b_r0, b_r1) -- their four fields are complete on r without any reduction -- and
    owns the x-slab [x_r0, x_r1) of the deposit grid for the gain update.  Per pass: (1) every rank sends, to each
    slab owner, its beams' fields over that slab (all-to-all, (world-1)/world of the rank's own 4 nb_r grids
    instead of an all-reduce of all 4 nb grids); (2) each rank updates the gain coefficient of ALL beams on its
    slab; (3) it sends every other rank the gain of that rank's beams over its slab (all-to-all, nb_r grids);
    (4) two scalars are all-reduced for the convergence measure.  At 256^3 / 60 beams / 8 ranks that is 3.6 GB +
    0.9 GB sent per rank and pass, against 58 GB of ring traffic per rank for the all-reduce of cbet_fixed_point.
    `engine`: begin(); field_passes_beams(use_gain, b0, b1) -> fields [4][nb][x][y][z]; update_gain_slab(fields, x0,
    x1) -> tensor {sum |dK|, sum |K|} over the slab; deposit_beams(b0, b1) -> beam_gain; attribute `gain`
    [nb][x][y][z].  The deposition grid is left un-reduced (allreduce_grid), as in cbet_fixed_point."""
    import torch.distributed as dist
    beams, slabs = _parts(nbeams, world_size), _parts(nx_halo, world_size)
    (b0, b1), (x0, x1) = beams[rank], slabs[rank]
    engine.begin()
    rep = {"passes": 0, "converged": False, "change": float("inf")}
    for it in range(gain_params.max_passes):
        fields = engine.field_passes_beams(it > 0, b0, b1)
        if world_size > 1:   # my beams' fields over slab s -> rank s; rank q's beams over my slab <- rank q
            _exchange(fields, lambda s: (slice(None), slice(b0, b1), slice(*slabs[s])),
                      lambda q: (slice(None), slice(*beams[q]), slice(x0, x1)), rank, world_size, group)
        ch = engine.update_gain_slab(fields, x0, x1)
        if world_size > 1:
            dist.all_reduce(ch, op=dist.ReduceOp.SUM, group=group)
            # the gain of rank r's beams over my slab -> rank r; my beams' gain over slab s <- rank s
            _exchange(engine.gain, lambda r: (slice(*beams[r]), slice(x0, x1)),
                      lambda s: (slice(b0, b1), slice(*slabs[s])), rank, world_size, group)
        ch = ch.cpu()
        rep["passes"] = it + 1
        rep["change"] = float(ch[0] / ch[1]) if float(ch[1]) > 0 else 0.0
        if rep["change"] < gain_params.tolerance:
            rep["converged"] = True
            break
    beam_gain = engine.deposit_beams(b0, b1)
    if world_size > 1:
        allreduce_grid(beam_gain, group)
    bg = beam_gain.cpu().numpy().copy()
    rep["beam_gain"] = bg
    rep["imbalance"] = float(abs(bg.sum()) / np.abs(bg).sum()) if np.abs(bg).sum() > 0 else 0.0
    return rep


def traced_pass(tracer, edep, rank=0, world_size=1, group=None, **launch_kw):
    """One full pass over the beams on `world_size` ranks: zero, trace this rank's share, combine."""
    edep.zero_()
    si, sc = shard_of_rank(rank, world_size)
    tracer.launch(edep, shard_index=si, shard_count=sc, **launch_kw)
    return allreduce_grid(edep, group)
