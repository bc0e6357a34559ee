"""RayTracer: torch-held device buffers around the C-ABI launch, and the multi-GPU step.

torch is plumbing here -- device memory, streams and torch.distributed (backend "nccl" = RCCL over
xGMI).  All computation happens in libcbet_mi355x.so through cbet_raytracing_3d_amd.api.

The multi-GPU scheme replaces main.cu:166-210 (/root/reference): instead of blocks of nbeams/nGPUs
whole beams per device (60/8 truncates to 7 and drops four beams) and a host-side sum of whole grids,
rank r traces the CONTIGUOUS part [T r / W, T (r+1) / W) of the beam-major list of T ray bundles into
its private (nx+2)(ny+2)(nz+2) grid, and the grids are combined by one reduce-scatter into x-slabs
(rank r ends up owning slab r of the sum: SweepPipeline, reduce_scatter_grid); allreduce_grid is kept
for callers that need the whole sum on every rank.
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
                    shard_count=1, ne3d=None, kappa3d=None, beam_lo=0, beam_hi=None, grid_beam0=0, grid_beams=0):
        """One trace with the CBET hooks on torch's current stream (node tables must be filled:
        tabulate(), or pass ne3d / kappa3d).  fields=False: deposit the absorbed energy into `out`
        ((n+2)^3 grid or a grid per beam); fields=True: the fused field pass, `out` = new_fields();
        fields="energy": the energy field alone, `out` = new_fields()[0] (a grid per beam).
        grid_beams > 0: the beam-resolved arrays (`out` when it is per beam, `gain`) hold only the grids of
        beams [grid_beam0, grid_beam0 + grid_beams) (cbet_params.grid_beam0 / grid_beams)."""
        ngrids = grid_beams if grid_beams > 0 else self.params.nbeams
        if fields == "energy":
            want = (ngrids,) + self.grid_shape
            per_beam = True
        elif fields:
            want = (4, ngrids) + self.grid_shape
            per_beam = True
        else:
            per_beam = out.dim() == 4
            want = ((ngrids,) + self.grid_shape) if per_beam else self.grid_shape
        if out.dtype != torch.float64 or not out.is_contiguous() or tuple(out.shape) != want:
            raise ValueError("out must be a contiguous float64 tensor of shape %s" % (want,))
        if gain is not None and (gain.dtype != torch.float64 or not gain.is_contiguous() or
                                 tuple(gain.shape) != (ngrids,) + self.grid_shape):
            raise ValueError("gain must be a contiguous float64 tensor of shape %s" % ((ngrids,) + self.grid_shape,))
        p = self.params.copy(per_beam_grids=1 if per_beam else 0, beam_lo=beam_lo,
                             beam_hi=self.params.nbeams if beam_hi is None else beam_hi,
                             shard_index=shard_index, shard_count=shard_count,
                             grid_beam0=grid_beam0, grid_beams=grid_beams)
        d = self.derived
        stream = torch.cuda.current_stream(self.device).cuda_stream
        quantity = api.DEPOSIT_FIELD_ENERGY if fields == "energy" else (api.DEPOSIT_FIELDS if fields else api.DEPOSIT_ENERGY)
        api.trace_cbet(0, d.nindices, ne3d, kappa3d, gain, quantity, out,
                       beam_gain, self.d_bbeam_norm, self.d_beam_norm, self.d_pow_r, self.d_phase_r, d.xconst,
                       d.yconst, d.zconst, p, gain_params, self.ctx, stream)
        return out

    def new_fields(self):
        """Zeroed [4][nbeams][(n+2)^3] field array (energy x path length, energy x displacement x/y/z)."""
        return torch.zeros((4, self.params.nbeams) + self.grid_shape, dtype=torch.float64, device=self.device)

    def gain_field(self, fields, gain, gain_params, change=None, ne3d=None, scratch=None, x_lo=0, x_hi=None, frozen=False,
                   pair_once=None):
        """Normalise `fields` in place and relax `gain` towards the gain coefficient they imply.
        pair_once: the kernel that evaluates each beam pair once, the cell's beams staged in LDS (True), or the ordered
        kernel in the CPU checker's sum order (False).  `scratch` is the C ABI's selector for the same choice (any
        tensor = pair-once; it is not touched) and is kept for callers of the earlier signature.
        x_lo, x_hi: only the planes [x_lo, x_hi) of the deposit grid (one rank's slab).
        frozen: fields[1:4] already hold k from an earlier call; only fields[0] is read and normalised."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if pair_once is None:
            pair_once = scratch is not None
        scratch = gain if pair_once else None
        api.gain_field_slab(fields, ne3d, gain, scratch, change, x_lo, self.grid_shape[0] if x_hi is None else x_hi,
                            self.params, _frozen(gain_params, frozen), self.ctx, stream)
        return gain

    def cbet_solve(self, edep, gain_params, rank=0, world_size=1, group=None, fields=None, gain=None, slabs=False,
                   force_collectives=False, sparse=False):
        """The CBET iteration, one rank's share (cbet_fixed_point -- or, with slabs=True, cbet_fixed_point_slabs,
        the exchange sized for xGMI -- with this device as the engine): the deposition pass is ADDED into
        `edep` (not reduced here: use allreduce_grid).  Single-rank callers can use the native loop instead:
        api.cbet_solve."""
        engine = _DeviceCbetEngine(self, edep, gain_params, fields, gain)
        engine.force_collectives = force_collectives
        if slabs:
            rep = cbet_fixed_point_slabs(engine, gain_params, self.params.nbeams, self.grid_shape[0], rank, world_size, group,
                                         sparse=sparse)
            rep["workspace_bytes"] = engine.slab_bytes()
            plan = engine.exchanger.plan
            rep["exchange"] = {"chunks": engine.exchanger.chunks, "bytes_sent": engine.exchanger.bytes_sent,
                               "staging_bytes": engine.exchanger.staging_bytes(),
                               "sparse": plan is not None,
                               "runs_per_exchange": plan.runs_out if plan is not None else None,
                               "dense_fraction": (8.0 * plan.runs_out / max(1, plan.dense_out)) if plan is not None else 1.0}
        else:
            rep = cbet_fixed_point(engine, gain_params, rank, world_size, group)
        rep["gain"] = engine.gain      # all beams (all-reduce loop) / this rank's beams (slab loop), whole grid
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
    """(shard_index, shard_count) of a rank: rank r traces the r-th contiguous 1/world_size of the bundle list."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside [0, world_size)")
    return rank, world_size


def allreduce_grid(edep, group=None, force=False):
    """Sum the per-rank deposition grids in place (RCCL all-reduce over xGMI with backend
    "nccl"; gloo on CPU tensors in the tests).  Replaces main.cu:178-210.  No-op without an
    initialised process group.  force: run the collective on a one-rank group too (RCCL smoke test)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        if edep.is_cuda and dist.get_backend(group) != "nccl":   # gloo has no device path: stage through the host
            host = edep.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            edep.copy_(host)
        else:
            dist.all_reduce(edep, op=dist.ReduceOp.SUM, group=group)
    return edep


def reduce_scatter_grid(grid, slab, group=None, async_op=False, force=False):
    """Combine the per-rank deposition grids so that rank r ends up with the SUM over ranks of x-slab r
    (`slab` = planes [r P/W, (r+1) P/W) of the plane-padded grid, P a multiple of the world size W): a
    reduce-scatter, half the xGMI traffic of the all-reduce and all a slab consumer (edepavg, a gain update, the
    host copy of a slab) needs.  RCCL with backend "nccl"; gloo (CPU tests) has no reduce-scatter for this
    layout, so there the grid is all-reduced and the slab copied out.  Returns the async work handle or None."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        slab.copy_(grid[: slab.shape[0]])
        return None
    if dist.get_backend(group) == "nccl":
        return dist.reduce_scatter_tensor(slab, grid, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    host = grid.cpu() if grid.is_cuda else grid      # gloo has no device path: stage through the host
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    r, pl = dist.get_rank(group), slab.shape[0]
    slab.copy_(host[r * pl:(r + 1) * pl])
    return None


class SweepPipeline:
    """Independent passes of the plain path on one rank, pipelined over HIP streams (the passes of a sweep do not
    feed each other: main.cu:96-232 run again on the same plasma).

    Per pass k, with two alternating buffer sets b = k % 2 (deposition grid, node tables + step records = a
    second context):
        prep stream b  : [tables b free = trace k-2 done, grid b free = combine k-2 done]  zero grid b,
                         tabulate the node tables, build the step records
        trace stream b : [prep k done]  trace this rank's share of the bundles into grid b, then enqueue the combine
        RCCL stream    : reduce-scatter of grid b over xGMI (torch's process-group stream, async)
    Nothing orders trace k+1 behind trace k (each buffer set has its own streams), so pass k+1's preparation AND the
    head of its trace run beside the drain of trace k -- a launch's last half millisecond runs at low occupancy, it
    cannot be shorter than one bundle's lifetime, and that is 0.8 ms of a 3.3 ms share at 8 ranks -- and combine k
    runs beside trace k+1.  Replaces
    the serial launch -> D2H -> host sum of main.cu:166-210.  The combined result of a pass is slab r of the
    grid on rank r (reduce_scatter_grid)."""

    def __init__(self, tracer, rank=0, world_size=1, group=None, overlap_traces=None, force_collectives=False):
        self.tr, self.rank, self.world, self.group = tracer, rank, world_size, group
        self.force = force_collectives      # run the RCCL combine on one rank too (smoke test of the collective path)
        # a rank's share of a sharded pass is a short launch whose drain is a quarter of it: overlap consecutive
        # traces there; a whole pass on one device gains 2 % and the kernel's own duration would no longer be
        # what the events around it measure, so it keeps one trace stream
        self.overlap_traces = (world_size > 1) if overlap_traces is None else bool(overlap_traces)
        p = tracer.params
        self.ctx = [tracer.ctx, api.Context(p, tracer.gpu)]
        planes = -(-(p.nx + 2) // world_size) * world_size          # padded to a multiple of the world size
        shape = (planes, p.ny + 2, p.nz + 2)
        dev = tracer.device
        self.grids = [torch.zeros(shape, dtype=torch.float64, device=dev) for _ in range(2)]
        self.slabs = [torch.zeros((planes // world_size,) + shape[1:], dtype=torch.float64, device=dev) for _ in range(2)]
        # one stream pair per buffer set: pass k+1 may start tracing while pass k is still draining
        self.s_prep = [torch.cuda.Stream(device=dev) for _ in range(2)]
        self.s_trace = [torch.cuda.Stream(device=dev) for _ in range(2)]
        if not self.overlap_traces:
            self.s_trace[1] = self.s_trace[0]
        self.ev_prep = [torch.cuda.Event() for _ in range(2)]
        self.ev_trace = [None, None]
        self.work = [None, None]
        self.kernel_events = []
        si, sc = shard_of_rank(rank, world_size)
        self.launch_p = p.copy(beam_lo=0, beam_hi=p.nbeams, shard_index=si, shard_count=sc)
        self.passes = 0

    def run_pass(self, timed=False):
        tr, d, b = self.tr, self.tr.derived, self.passes % 2
        self.passes += 1
        with torch.cuda.stream(self.s_prep[b]):
            sp = self.s_prep[b].cuda_stream
            if self.ev_trace[b] is not None:
                self.s_prep[b].wait_event(self.ev_trace[b])
            if self.work[b] is not None:
                self.work[b].wait()            # this stream waits for combine k-2 before the grid is cleared
                self.work[b] = None
            self.grids[b].zero_()
            api.tabulate_plasma(self.ctx[b], self.launch_p, tr.d_te, tr.d_r, tr.d_ne, sp)
            api.prepare_step_records(self.ctx[b], self.launch_p, None, None, d.xconst, d.yconst, d.zconst, sp)
            self.ev_prep[b].record()
        with torch.cuda.stream(self.s_trace[b]):
            st = self.s_trace[b].cuda_stream
            self.s_trace[b].wait_event(self.ev_prep[b])
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            api.trace_nodes(0, d.nindices, None, None, self.grids[b], tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r,
                            tr.d_phase_r, d.xconst, d.yconst, d.zconst, self.launch_p, self.ctx[b], st)
            if timed:
                e1.record()
                self.kernel_events.append((e0, e1))
            self.work[b] = reduce_scatter_grid(self.grids[b], self.slabs[b], self.group, async_op=True, force=self.force)
            # "grid b may be cleared again": recorded AFTER the combine was enqueued -- on one rank (and with gloo) the
            # combine is a copy on this very stream, and pass k+2's grid.zero_() must not overtake it; with RCCL the
            # collective runs on the process group's stream and is waited for through its work handle
            self.ev_trace[b] = torch.cuda.Event()
            self.ev_trace[b].record()
        return b

    def time_trace_alone(self, reps=3):
        """Average duration (seconds) of this rank's trace launch when NOTHING else runs beside it: with more than one
        rank the pipeline lets consecutive passes' trace kernels overlap, so the events around a launch there measure a
        stretched duration; this is the time the launch needs (what a roofline fraction has to be priced with)."""
        tr, d = self.tr, self.tr.derived
        self.finish()
        times = []
        with torch.cuda.stream(self.s_trace[0]):
            st = self.s_trace[0].cuda_stream
            for _ in range(reps + 1):
                self.grids[0].zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                api.trace_nodes(0, d.nindices, None, None, self.grids[0], tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r,
                                tr.d_phase_r, d.xconst, d.yconst, d.zconst, self.launch_p, self.ctx[0], st)
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1) * 1e-3)
        torch.cuda.synchronize(tr.device)
        return sum(times[1:]) / reps

    def warm(self):
        """Run the combine once on the (zero) buffers: RCCL builds its communicator, channels and staging buffers on
        the first collective of a kind -- set-up, like the reference's cudaMalloc in its Init phase (main.cu:131-152),
        not part of a pass.  No-op on one rank."""
        if self.world > 1 or self.force:
            for b in range(2):
                w = reduce_scatter_grid(self.grids[b], self.slabs[b], self.group, async_op=True, force=self.force)
                if w is not None:
                    w.wait()
            torch.cuda.synchronize(self.tr.device)

    def finish(self):
        """Wait for everything in flight; returns the slab of the last pass (this rank's planes of the sum)."""
        for b in range(2):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None
        torch.cuda.synchronize(self.tr.device)
        return self.slabs[(self.passes - 1) % 2] if self.passes else None

    def counters(self, reset=False):
        stream = torch.cuda.current_stream(self.tr.device).cuda_stream
        c0, c1 = self.ctx[0].counters(stream, reset), self.ctx[1].counters(stream, reset)
        for name, _ in api.Counters._fields_:
            setattr(c0, name, getattr(c0, name) + getattr(c1, name))
        return c0

    def close(self):
        self.finish()
        self.ctx[1].close()


class _DeviceCbetEngine:
    """The per-rank compute of the CBET iteration on a RayTracer's device (see cbet_fixed_point and
    cbet_fixed_point_slabs).  The all-reduce loop keeps every beam's arrays over the whole grid
    (cbet_cbet_workspace_bytes: 41.2 GB at 256^3 / 60 beams); the slab-owned loop allocates, in begin_slabs, only
    its own beams over the whole grid and all beams over its own x-slab (cbet_cbet_slab_workspace_bytes)."""

    def __init__(self, tracer, edep, gain_params, fields=None, gain=None):
        self.tr, self.edep, self.gp = tracer, edep, gain_params
        self._fields, self._gain = fields, gain
        self.change = torch.zeros(2, dtype=torch.float64, device=tracer.device)
        self.beam_gain = torch.zeros(tracer.params.nbeams, dtype=torch.float64, device=tracer.device)

    # ---- all-reduce loop: whole-grid arrays of every beam
    def begin(self):
        tr = self.tr
        self.fields = tr.new_fields() if self._fields is None else self._fields
        self.gain = tr.new_grid(per_beam=True) if self._gain is None else self._gain
        tr.tabulate()
        self.gain.zero_()

    def field_passes(self, use_gain, shard_index, shard_count, full=True):
        """full: all four fields; else the energy field (fields[0]) alone -- fields[1:4] keep the k of the last full pass."""
        (self.fields if full else self.fields[0]).zero_()
        self.tr.launch_cbet(self.fields if full else self.fields[0], self.gp, fields=True if full else "energy",
                            gain=self.gain if use_gain else None, shard_index=shard_index, shard_count=shard_count)
        return self.fields

    def update_gain(self, fields, frozen=False):
        self.change.zero_()
        self.tr.gain_field(fields, self.gain, self.gp, self.change, pair_once=True, frozen=frozen)
        return self.change

    def deposit(self, shard_index, shard_count):
        self.beam_gain.zero_()
        self.tr.launch_cbet(self.edep, self.gp, gain=self.gain, beam_gain=self.beam_gain,
                            shard_index=shard_index, shard_count=shard_count)
        return self.beam_gain

    # ---- slab-owned loop: own beams [b0, b1) over the whole grid, all beams over the own planes [x0, x1)
    def begin_slabs(self, b0, b1, x0, x1):
        tr, dev = self.tr, self.tr.device
        nb, gs = tr.params.nbeams, tr.grid_shape
        self.b0, self.b1, self.x0, self.x1 = b0, b1, x0, x1
        f64 = dict(dtype=torch.float64, device=dev)
        self.own_fields = torch.zeros((4, b1 - b0) + gs, **f64)
        self.gain_own = torch.zeros((b1 - b0,) + gs, **f64)
        self.slab_fields = torch.zeros((4, nb, x1 - x0) + gs[1:], **f64)
        self.gain_slab = torch.zeros((nb, x1 - x0) + gs[1:], **f64)
        self.gain = self.gain_own            # what a caller gets back: this rank's beams over the whole grid
        tr.tabulate()

    def slab_bytes(self):
        """Device bytes this rank's slab loop holds: the arrays of begin_slabs, the exchange's two staging buffers and
        (sparse exchanges) the segment lists."""
        arrays = 8 * sum(t.numel() for t in (self.own_fields, self.gain_own, self.slab_fields, self.gain_slab))
        xch = getattr(self, "exchanger", None)
        if xch is None:
            return arrays
        plan = getattr(xch, "plan", None)
        return arrays + xch.staging_bytes() + (plan.list_bytes() if plan is not None else 0)

    def support_mask(self):
        """bool [own beams][X][Y][Z]: every node this rank's beams can ever deposit into -- their rays traced in the
        reference's bookkeeping mode (absorption = 0, def.cuh:118: the energy never decays, so no ray stops before it
        leaves the grid or runs out of steps) into beam-resolved grids.  Ray paths do not depend on the gain, so this
        footprint contains the footprint of every pass of the iteration.  Uses own_fields[1:] as scratch (call it
        before the first field pass)."""
        tr, nbr = self.tr, self.b1 - self.b0
        if nbr == 0:
            return torch.zeros((0,) + tr.grid_shape, dtype=torch.bool, device=tr.device)
        d = tr.derived
        p = tr.params.copy(absorption=0, per_beam_grids=1, beam_lo=self.b0, beam_hi=self.b1, grid_beam0=self.b0, grid_beams=nbr)
        tmp = self.own_fields[1]
        tmp.zero_()
        stream = torch.cuda.current_stream(tr.device).cuda_stream
        api.trace_nodes(0, d.nindices, None, None, tmp, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                        d.xconst, d.yconst, d.zconst, p, tr.ctx, stream)
        mask = tmp != 0
        tmp.zero_()
        tr.counters(reset=True)       # the footprint pass is set-up, not part of the iteration's ray-step count
        return mask

    def field_passes_beams(self, use_gain, full=True):
        out = self.own_fields if full else self.own_fields[0]
        out.zero_()
        if self.b1 > self.b0:
            self.tr.launch_cbet(out, self.gp, fields=True if full else "energy", gain=self.gain_own if use_gain else None,
                                beam_lo=self.b0, beam_hi=self.b1, grid_beam0=self.b0, grid_beams=self.b1 - self.b0)
        return self.own_fields

    def update_gain_slab(self, frozen=False):
        self.change.zero_()
        if self.x1 > self.x0:
            stream = torch.cuda.current_stream(self.tr.device).cuda_stream
            api.gain_field_packed(self.slab_fields, None, self.gain_slab, self.gain_slab, self.change, self.x0, self.x1,
                                  self.tr.params, _frozen(self.gp, frozen), self.tr.ctx, stream)
        return self.change

    def deposit_beams(self):
        self.beam_gain.zero_()
        if self.b1 > self.b0:
            self.tr.launch_cbet(self.edep, self.gp, gain=self.gain_own, beam_gain=self.beam_gain, beam_lo=self.b0,
                                beam_hi=self.b1, grid_beam0=self.b0, grid_beams=self.b1 - self.b0)
        return self.beam_gain


def _frozen(gain_params, frozen):
    """gain_params with directions_frozen set as asked (a copy when it has to change)."""
    if bool(gain_params.directions_frozen) == bool(frozen):
        return gain_params
    g = type(gain_params).from_buffer_copy(gain_params)
    g.directions_frozen = 1 if frozen else 0
    return g


def _agree(t, group, world_size):
    """Make a small per-rank tensor that steers control flow identical on all ranks (rank 0's copy): collectives
    need not return bit-identical values everywhere, and a stop decision taken from slightly different numbers
    would leave ranks waiting in different collectives."""
    import torch.distributed as dist
    if world_size > 1:
        host = t.detach().cpu()
        dist.broadcast(host, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
        return host
    return t.detach().cpu()


def cbet_fixed_point(engine, gain_params, rank=0, world_size=1, group=None):
    """The CBET fixed-point iteration over `world_size` ranks (parity unpinned).

    Every pass: each rank deposits the fields of ITS share of the ray bundles (the plain pass's sharding) -- all four
    in the first gain_params.direction_passes passes, which build the direction field k, the energy field alone
    afterwards (gain changes ray energies, not ray paths: k of the gain-free pass is kept) --, the deposited fields
    are summed over ranks with one all-reduce, and every rank updates the full gain coefficient from them
    (redundantly -- it needs all of it for its own rays).
    Stops when sum |dK| / sum |K| < tolerance (rank 0's value, broadcast), then runs the deposition pass and
    all-reduces the per-beam energy balance.  `engine` supplies the per-rank compute:
        begin(); field_passes(use_gain, shard_index, shard_count, full) -> fields tensor;
        update_gain(fields, frozen) -> tensor {sum |dK|, sum |K|}; deposit(shard_index, shard_count) -> beam_gain tensor
    (the device engine is RayTracer.cbet_solve's; the CPU tests drive this loop with an oracle engine).
    Returns {passes, converged, change, beam_gain, imbalance}."""
    si, sc = shard_of_rank(rank, world_size)
    force = getattr(engine, "force_collectives", False)   # one rank, but every collective really runs (RCCL smoke test)
    engine.begin()
    rep = {"passes": 0, "converged": False, "change": float("inf")}
    for it in range(gain_params.max_passes):
        full = it < gain_params.direction_passes
        fields = engine.field_passes(it > 0, si, sc, full)
        if world_size > 1 or force:
            allreduce_grid(fields if full else fields[0], group, force)
        ch = _agree(engine.update_gain(fields, not full), group, world_size)
        rep["passes"] = it + 1
        rep["change"] = float(ch[0] / ch[1]) if float(ch[1]) > 0 else 0.0
        if rep["change"] < gain_params.tolerance:
            rep["converged"] = True
            break
    beam_gain = engine.deposit(si, sc)
    if world_size > 1 or force:
        allreduce_grid(beam_gain, group, force)
    bg = beam_gain.cpu().numpy().copy()
    rep["beam_gain"] = bg
    rep["imbalance"] = float(abs(bg.sum()) / np.abs(bg).sum()) if np.abs(bg).sum() > 0 else 0.0
    return rep


def _parts(total, world_size):
    """Contiguous near-equal parts of range(total): [(lo, hi)] per rank (cbet_cbet_slab_workspace_bytes uses the
    same rule).  Parts are empty when world_size > total."""
    return [((r * total) // world_size, ((r + 1) * total) // world_size) for r in range(world_size)]


def exchange_staging_elems(nbeams, nx_halo, plane, world_size, force=False):
    """Doubles in ONE staging buffer of the slab loop's exchanges (there are two: send and receive): the largest
    chunk any exchange moves at once = the most beams a rank owns x the most planes a rank owns x one plane, for
    one field component (cbet_cbet_slab_workspace_bytes counts 2 x this).  0 on one rank (nothing is exchanged)."""
    if world_size <= 1 and not force:
        return 0
    return (-(-nbeams // world_size)) * (-(-nx_halo // world_size)) * plane


class _Exchanger:
    """The all-to-all exchanges of the slab-owned CBET loop over point-to-point xGMI links (RCCL send/recv; gloo in the
    CPU tests), stream-ordered and chunked:

    * no host synchronisation with RCCL: the producer's stream records an event, everything below runs on a
      communication stream that waits for it, and the consumer's stream waits for the event recorded at the end;
    * one peer pair at a time, in W - 1 rounds: in round k rank r sends to r + k and receives from r - k (every link
      of the point-to-point fabric carries one message per round, no rank is the target of two senders), and a
      multi-component array moves one component at a time: the pack copy, the grouped send/recv and the unpack copy
      of a chunk reuse ONE send and ONE receive staging buffer (sized by exchange_staging_elems, allocated once and
      counted in cbet_cbet_slab_workspace_bytes) instead of a contiguous copy per peer all at once;
    * with a backend that has no device path (gloo) device tensors are staged through the host, chunk by chunk.

    rank r sends src[send_index(s)] to every other rank s and stores what s sends it in dst[recv_index(s)]; its own
    part is copied.  Index tuples select strided views, the first index being the component axis when
    `components` is true.  Empty parts (a rank without beams or planes) are skipped on both sides."""

    def __init__(self, device, staging_elems, group=None, force_collectives=False):
        import torch.distributed as dist
        self.group, self.device = group, torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.nccl = self.cuda and dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl"
        self.force = force_collectives      # run the send/recv machinery on one rank too (a self-exchange: RCCL smoke test)
        self.stream = torch.cuda.Stream(device=self.device) if self.nccl else None
        stage_dev = self.device if (self.nccl or not self.cuda) else torch.device("cpu")
        self.send_buf = torch.empty(staging_elems, dtype=torch.float64, device=stage_dev)
        self.recv_buf = torch.empty(staging_elems, dtype=torch.float64, device=stage_dev)
        self.chunks = 0                     # chunks moved so far (tests, logs)
        self.bytes_sent = 0

    def staging_bytes(self):
        return 8 * (self.send_buf.numel() + self.recv_buf.numel()) if self.send_buf.device == self.device else 0

    def _chunks(self, view, components):
        if view.numel() == 0:
            return []
        return [view[c] for c in range(view.shape[0])] if components else [view]

    def run(self, src, send_index, dst, recv_index, rank, world_size, components=False):
        import torch.distributed as dist
        dst[recv_index(rank)] = src[send_index(rank)]
        if world_size == 1 and not self.force:
            return
        producer = torch.cuda.current_stream(self.device) if self.nccl else None
        if self.nccl:
            ev = torch.cuda.Event()
            ev.record(producer)
            self.stream.wait_event(ev)
        ctx = torch.cuda.stream(self.stream) if self.nccl else _NullContext()
        with ctx:
            rounds = range(1, world_size) if world_size > 1 else [0]      # [0]: the forced self-exchange of one rank
            for k in rounds:
                to, frm = (rank + k) % world_size, (rank - k) % world_size
                outs = self._chunks(src[send_index(to)], components)
                wants = self._chunks(dst[recv_index(frm)], components)
                for c in range(max(len(outs), len(wants))):
                    ops = []
                    if c < len(outs):
                        n = outs[c].numel()
                        sb = self.send_buf[:n].view(outs[c].shape)
                        sb.copy_(outs[c])                                # pack (device: on the communication stream)
                        peer = to if self.group is None else dist.get_global_rank(self.group, to)
                        ops.append(dist.P2POp(dist.isend, sb, peer, self.group))
                        self.bytes_sent += 8 * n
                    if c < len(wants):
                        m = wants[c].numel()
                        rb = self.recv_buf[:m].view(wants[c].shape)
                        peer = frm if self.group is None else dist.get_global_rank(self.group, frm)
                        ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
                    if ops:
                        for req in dist.batch_isend_irecv(ops):
                            req.wait()     # RCCL: the communication stream waits (no host block); gloo: the host waits
                        self.chunks += 1
                    if c < len(wants):
                        wants[c].copy_(rb)                               # unpack
        if self.nccl:
            done = torch.cuda.Event()
            done.record(self.stream)
            producer.wait_event(done)

    def _pack(self, arr, stride, hy, hz, seg, out):
        n = seg.shape[0]
        if arr.is_cuda:
            api.pack_segments(arr, stride, hy, hz, seg, n, out, torch.cuda.current_stream(arr.device).cuda_stream)
        else:
            idx, valid = _pack_rows_cpu(arr, stride, hz, seg)
            out[: 8 * n].view(n, 8).copy_(arr.reshape(-1)[idx] * valid)

    def _unpack(self, arr, stride, hy, hz, seg, buf):
        n = seg.shape[0]
        if arr.is_cuda:
            api.unpack_segments(arr, stride, hy, hz, seg, n, buf, torch.cuda.current_stream(arr.device).cuda_stream)
        else:
            idx, valid = _pack_rows_cpu(arr, stride, hz, seg)
            arr.view(-1)[idx[valid]] = buf[: 8 * n].view(n, 8)[valid]

    def run_sparse(self, src, send_index, dst, recv_index, plan, to_slabs, rank, world_size, ncomp=0):
        """The same exchange, moving only the 64-byte z-runs of `plan` (SegmentPlan): to_slabs = exchange 1 (my beams'
        fields to the slab owners: pack from my whole-grid array, unpack into my slab array), else exchange 2 (the gain of
        the peers' beams over my slab back to them).  ncomp > 0: the arrays carry that many leading components, one message
        each.  One peer pair per round, every message through the two staging buffers, stream-ordered as in run()."""
        import torch.distributed as dist
        dst[recv_index(rank)] = src[send_index(rank)]       # the own part: a dense local copy
        if world_size == 1 and not self.force:
            return
        hy, hz = plan.Y, plan.Z
        out_lists, in_lists = (plan.own_side, plan.slab_side) if to_slabs else (plan.slab_side, plan.own_side)
        out_stride, in_stride = (plan.own_stride, plan.slab_stride) if to_slabs else (plan.slab_stride, plan.own_stride)
        dev_stage = self.send_buf.device == src.device
        producer = torch.cuda.current_stream(self.device) if self.nccl else None
        if self.nccl:
            ev = torch.cuda.Event()
            ev.record(producer)
            self.stream.wait_event(ev)
        with (torch.cuda.stream(self.stream) if self.nccl else _NullContext()):
            rounds = range(1, world_size) if world_size > 1 else [0]
            for k in rounds:
                to, frm = (rank + k) % world_size, (rank - k) % world_size
                seg_out, seg_in = out_lists[to], in_lists[frm]
                n_out, n_in = seg_out.shape[0], seg_in.shape[0]
                for c in range(max(1, ncomp)):
                    s_arr = src[c] if ncomp else src
                    d_arr = dst[c] if ncomp else dst
                    ops = []
                    if n_out:
                        if dev_stage:
                            self._pack(s_arr, out_stride, hy, hz, seg_out, self.send_buf)
                            sb = self.send_buf[: 8 * n_out]
                        else:                   # gloo with device arrays: pack on the device, stage through the host
                            tmp = torch.empty(8 * n_out, dtype=torch.float64, device=src.device)
                            self._pack(s_arr, out_stride, hy, hz, seg_out, tmp)
                            sb = self.send_buf[: 8 * n_out]
                            sb.copy_(tmp)
                        peer = to if self.group is None else dist.get_global_rank(self.group, to)
                        ops.append(dist.P2POp(dist.isend, sb, peer, self.group))
                        self.bytes_sent += 64 * n_out
                    if n_in:
                        rb = self.recv_buf[: 8 * n_in]
                        peer = frm if self.group is None else dist.get_global_rank(self.group, frm)
                        ops.append(dist.P2POp(dist.irecv, rb, peer, self.group))
                    if ops:
                        for req in dist.batch_isend_irecv(ops):
                            req.wait()
                        self.chunks += 1
                    if n_in:
                        self._unpack(d_arr, in_stride, hy, hz, seg_in, rb if dev_stage else rb.to(dst.device))
        if self.nccl:
            done = torch.cuda.Event()
            done.record(self.stream)
            producer.wait_event(done)


def _segment_rows(support, x0, x1):
    """Rows (beam, x - x0, y, z // 8) of the 64-byte z-runs of planes [x0, x1) in which `support` (bool
    [beams][X][Y][Z]) is set anywhere: the unit of the sparse exchange (cbet_pack_segments)."""
    nb, X, Y, Z = support.shape
    zs = (Z + 7) // 8
    sub = support[:, x0:x1]
    if zs * 8 != Z:
        sub = torch.nn.functional.pad(sub, (0, zs * 8 - Z))
    return sub.reshape(nb, x1 - x0, Y, zs, 8).any(-1).nonzero().to(torch.int32)


def _pack_rows_cpu(src, beam_stride, hz, seg):
    """torch restatement of cbet_pack_segments for host tensors (the gloo tests); returns (values [n][8], flat index, valid)"""
    zsegs = (hz + 7) // 8
    rows, run = seg[:, 0].long(), seg[:, 1].long()
    z = 8 * (run % zsegs)[:, None] + torch.arange(8)
    valid = z < hz
    idx = rows[:, None] * beam_stride + (run // zsegs)[:, None] * hz + z.clamp(max=hz - 1)
    return idx, valid


class SegmentPlan:
    """Who sends which 64-byte z-runs to whom in the slab-owned CBET loop, fixed for the life of a solve.

    `support` [own beams][X][Y][Z] marks every node this rank's beams can EVER deposit into -- the footprint of their
    rays traced to the exit of the grid whatever their energy (ray paths do not depend on the gain; which step a ray is
    absorbed at does) -- so the lists hold every entry any pass can make non-zero, and every entry of a beam's gain
    coefficient its rays can read.  For each peer s the rank keeps the runs of its beams inside slab s (what it packs
    for exchange 1 and unpacks in exchange 2), and -- received from the peers once, by send/recv -- the runs of every
    peer q's beams inside its own slab (what it unpacks in exchange 1 and packs for exchange 2)."""

    def __init__(self, support, beams, slabs, rank, world_size, group, device):
        import torch.distributed as dist
        nbr, X, Y, Z = support.shape
        self.Y, self.Z, self.zsegs = Y, Z, (Z + 7) // 8
        self.own_stride, self.slab_planes = X * Y * Z, slabs[rank][1] - slabs[rank][0]
        self.slab_stride = self.slab_planes * Y * Z
        mine = []            # per peer s: rows (b_local, x_rel, y, zs) of my beams in slab s
        for s in range(world_size):
            mine.append(_segment_rows(support, *slabs[s]).cpu())
        # the peers' rows for my slab: counts first, then the lists, point to point
        theirs = [None] * world_size
        theirs[rank] = mine[rank]
        if world_size > 1:
            cuda_nccl = dist.get_backend(group) == "nccl"
            cdev = device if cuda_nccl else "cpu"
            counts = torch.tensor([m.shape[0] for m in mine], dtype=torch.int64, device=cdev)
            allc = [torch.zeros_like(counts) for _ in range(world_size)]
            dist.all_gather(allc, counts, group=group)
            for k in range(1, world_size):
                to, frm = (rank + k) % world_size, (rank - k) % world_size
                ops, rb = [], None
                peer = lambda r_: r_ if group is None else dist.get_global_rank(group, r_)
                if mine[to].shape[0]:
                    ops.append(dist.P2POp(dist.isend, mine[to].to(cdev).contiguous(), peer(to), group))
                n_in = int(allc[frm][rank])
                if n_in:
                    rb = torch.empty((n_in, 4), dtype=torch.int32, device=cdev)
                    ops.append(dist.P2POp(dist.irecv, rb, peer(frm), group))
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                if cuda_nccl:
                    torch.cuda.synchronize(device)
                theirs[frm] = rb.cpu() if rb is not None else torch.zeros((0, 4), dtype=torch.int32)
        zs = self.zsegs

        def pairs(rows, beam_offset, x_offset):
            if rows.shape[0] == 0:
                return torch.zeros((0, 2), dtype=torch.int32, device=device)
            r = rows.long()
            out = torch.stack([r[:, 0] + beam_offset, ((r[:, 1] + x_offset) * Y + r[:, 2]) * zs + r[:, 3]], 1)
            return out.to(torch.int32).contiguous().to(device)
        # what I address in MY whole-grid arrays (own_fields, gain_own): my beams, absolute planes, per peer slab
        self.own_side = [pairs(mine[s], 0, slabs[s][0]) for s in range(world_size)]
        # what I address in MY slab arrays (slab_fields, gain_slab): peer q's beams (global row), planes relative to my slab
        self.slab_side = [pairs(theirs[q], beams[q][0], 0) for q in range(world_size)]
        solo = world_size == 1          # the forced self-exchange of a one-rank group moves the rank's own part
        self.max_out = max([t.shape[0] for i, t in enumerate(self.own_side) if i != rank or solo] + [0])
        self.max_in = max([t.shape[0] for i, t in enumerate(self.slab_side) if i != rank or solo] + [0])
        self.runs_out = sum(t.shape[0] for i, t in enumerate(self.own_side) if i != rank)    # exchange 1 sends, exchange 2 receives
        self.runs_in = sum(t.shape[0] for i, t in enumerate(self.slab_side) if i != rank)    # exchange 1 receives, exchange 2 sends
        self.dense_out = nbr * (X - self.slab_planes) * Y * Z      # doubles a dense exchange would send

    def staging_elems(self):
        return 8 * max(self.max_out, self.max_in)

    def list_bytes(self):
        return 8 * (sum(t.shape[0] for t in self.own_side) + sum(t.shape[0] for t in self.slab_side))


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def cbet_fixed_point_slabs(engine, gain_params, nbeams, nx_halo, rank=0, world_size=1, group=None, sparse=False):
    """The CBET fixed-point iteration with storage and exchange sized for 8 ranks on point-to-point xGMI (SURVEY
    8(f) f1; parity unpinned; same passes and same result as cbet_fixed_point).

    Rank r traces WHOLE beams [b_r0, b_r1) -- their four fields are complete on r without any reduction -- and
    owns the x-slab [x_r0, x_r1) of the deposit grid for the gain update.  It STORES only
        its own beams over the whole grid : own_fields [4][nb_r][X][Y][Z], gain_own [nb_r][X][Y][Z]
        all beams over its own slab       : slab_fields [4][nb][x_r][Y][Z], gain_slab [nb][x_r][Y][Z]
    i.e. (5 nb_r + 6 nb / W) grids instead of 6 nb: 92 GB per rank at 512^3 / 60 beams / 8 ranks against 391 GB
    (cbet_cbet_slab_workspace_bytes).  Per pass: (1) every rank sends each slab owner its beams' fields over that
    slab (all-to-all, (W-1)/W of own_fields); (2) each rank updates the gain of ALL beams on its slab; (3) it sends
    every rank the gain of that rank's beams over its slab (all-to-all, gain_slab); (4) two scalars are
    all-reduced for the convergence measure and rank 0's copy decides.  At 256^3 / 60 beams / 8 ranks that is
    3.6 GB + 0.9 GB sent per rank in a direction-building pass (the first) and 0.9 GB + 0.9 GB in every later one
    (energy field only), against 58 / 14 GB of ring traffic per rank for the all-reduce loop.
    `engine`: begin_slabs(b0, b1, x0, x1); field_passes_beams(use_gain, full) -> own_fields; attributes slab_fields,
    gain_slab, gain_own; update_gain_slab(frozen) -> tensor {sum |dK|, sum |K|} over the slab; deposit_beams() ->
    beam_gain.  The deposition grid is left un-reduced (allreduce_grid / reduce_scatter_grid)."""
    import torch.distributed as dist
    beams, slabs = _parts(nbeams, world_size), _parts(nx_halo, world_size)
    (b0, b1), (x0, x1) = beams[rank], slabs[rank]
    engine.begin_slabs(b0, b1, x0, x1)
    plane = int(engine.slab_fields.shape[-1] * engine.slab_fields.shape[-2])
    force = getattr(engine, "force_collectives", False)   # one rank, but every collective really runs (RCCL smoke test)
    # sparse = True: the exchanges move only the 64-byte z-runs a beam's rays can ever touch (SegmentPlan) instead of
    # dense sub-arrays.  Exact, but it does not pay for this physics: traced to the exit of the grid whatever their energy
    # -- the only footprint that is guaranteed to contain every pass's -- the refracted rays of an OMEGA beam visit 73 %
    # of the nodes of the 256^3 grid (83 % of its z-runs), and the pack / unpack kernels of 0.8 GB take longer (0.84 ms per
    # exchange) than the dense exchange's strided copies (0.57 ms): profiles/r3/cbet_rank_share.log.  Kept as an option
    # for plasmas / beam sets whose footprint is small.
    support = engine.support_mask() if (sparse and hasattr(engine, "support_mask") and (world_size > 1 or force)) else None
    plan = SegmentPlan(support, beams, slabs, rank, world_size, group, engine.slab_fields.device) if support is not None else None
    del support
    staging = plan.staging_elems() if plan is not None else exchange_staging_elems(nbeams, nx_halo, plane, world_size, force)
    xch = _Exchanger(engine.slab_fields.device, staging, group, force_collectives=force)
    xch.plan = plan
    engine.exchanger = xch
    rep = {"passes": 0, "converged": False, "change": float("inf")}
    for it in range(gain_params.max_passes):
        full = it < gain_params.direction_passes
        comps = slice(None) if full else slice(0, 1)    # after the direction-building passes only the energy field moves
        own = engine.field_passes_beams(it > 0, full)
        # my beams' fields over slab s -> rank s; rank q's beams over my slab <- rank q
        if plan is not None:
            xch.run_sparse(own, lambda s: (comps, slice(None), slice(*slabs[s])),
                           engine.slab_fields, lambda q: (comps, slice(*beams[q])), plan, True, rank, world_size,
                           ncomp=4 if full else 1)
        else:
            xch.run(own, lambda s: (comps, slice(None), slice(*slabs[s])),
                    engine.slab_fields, lambda q: (comps, slice(*beams[q])), rank, world_size, components=True)
        ch = engine.update_gain_slab(not full)
        if world_size > 1 or force:
            if ch.is_cuda and dist.get_backend(group) != "nccl":
                host = ch.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                ch = host
            else:
                dist.all_reduce(ch, op=dist.ReduceOp.SUM, group=group)
        # the gain of rank q's beams over my slab -> rank q; my beams' gain over slab s <- rank s
        if plan is not None:
            xch.run_sparse(engine.gain_slab, lambda q: (slice(*beams[q]),),
                           engine.gain_own, lambda s: (slice(None), slice(*slabs[s])), plan, False, rank, world_size)
        else:
            xch.run(engine.gain_slab, lambda q: (slice(*beams[q]),),
                    engine.gain_own, lambda s: (slice(None), slice(*slabs[s])), rank, world_size)
        ch = _agree(ch, group, world_size)
        rep["passes"] = it + 1
        rep["change"] = float(ch[0] / ch[1]) if float(ch[1]) > 0 else 0.0
        if rep["change"] < gain_params.tolerance:
            rep["converged"] = True
            break
    beam_gain = engine.deposit_beams()
    if world_size > 1 or force:
        allreduce_grid(beam_gain, group, force)
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
