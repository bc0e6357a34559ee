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
import os

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

    def new_grid(self, per_beam=False, zpitch=None):
        """A zeroed deposition grid; per_beam=True: one grid per beam (cbet_params.per_beam_grids).  zpitch: rows of that
        many doubles (>= nz + 2; True = the next multiple of 8: whole 64-byte lines) instead of the reference's dense
        rows -- launch() recognises such a grid by its shape (cbet_params.edep_zpitch); [..., :nz + 2] is the reference's
        view of it.  Plain path only."""
        if zpitch:
            if per_beam:
                raise ValueError("a padded row pitch applies to the plain path's single grid")
            zp = -(-self.grid_shape[2] // 8) * 8 if zpitch is True else int(zpitch)
            return torch.zeros(self.grid_shape[:2] + (zp,), dtype=torch.float64, device=self.device)
        shape = ((self.params.nbeams,) + self.grid_shape) if per_beam else self.grid_shape
        return torch.zeros(shape, dtype=torch.float64, device=self.device)

    def launch(self, edep, shard_index=0, shard_count=1, beam_lo=0, beam_hi=None,
               kernel_variant=None, force_wide_index=None, use_host_trig=True, stats=None):
        """Enqueue one launch_ray_XYZ on torch's current stream, accumulating into `edep`.  stats=True: the launch also
        counts the deposit windows' diagnostics (cbet_params.window_stats; a timed launch does not)."""
        per_beam = edep.dim() == 4
        want = ((self.params.nbeams,) + self.grid_shape) if per_beam else self.grid_shape
        # a single grid whose rows are longer than nz + 2 is a padded grid (new_grid(zpitch=...))
        padded = (not per_beam) and edep.dim() == 3 and tuple(edep.shape[:2]) == want[:2] and edep.shape[2] > want[2]
        if edep.dtype != torch.float64 or not edep.is_contiguous() or (tuple(edep.shape) != want and not padded):
            raise ValueError("edep must be a contiguous float64 tensor of shape %s (or nbeams x that, or with padded rows)" % (self.grid_shape,))
        p = self.params.copy(per_beam_grids=1 if per_beam else 0, beam_lo=beam_lo,
                             beam_hi=self.params.nbeams if beam_hi is None else beam_hi,
                             shard_index=shard_index, shard_count=shard_count, edep_zpitch=int(edep.shape[2]) if padded else 0)
        if kernel_variant is not None:
            p.kernel_variant = kernel_variant
        if force_wide_index is not None:
            p.force_wide_index = force_wide_index
        if stats is not None:
            p.window_stats = 1 if stats else 0
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
                   force_collectives=False, sparse=False, **slab_options):
        """The CBET iteration, one rank's share (cbet_fixed_point -- or, with slabs=True, cbet_fixed_point_slabs,
        the exchange sized for xGMI -- with this device as the engine): the deposition pass is ADDED into
        `edep` (not reduced here: use allreduce_grid).  Single-rank callers can use the native loop instead:
        api.cbet_solve."""
        engine = _DeviceCbetEngine(self, edep, gain_params, fields, gain)
        engine.force_collectives = force_collectives
        if slabs:
            rep = cbet_fixed_point_slabs(engine, gain_params, self.params.nbeams, self.grid_shape[0], rank, world_size, group,
                                         sparse=sparse, **slab_options)      # trace_groups=, slab_layout=
            rep["workspace_bytes"] = engine.slab_bytes()
            plan = engine.exchanger.plan
            rep["exchange"] = {"chunks": engine.exchanger.chunks, "messages": engine.exchanger.messages, "bytes_sent": engine.exchanger.bytes_sent,
                               "staging_bytes": engine.exchanger.staging_bytes(),
                               "sparse": plan is not None, "two_channels": engine.exchanger.two_channels,
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

    def __init__(self, tracer, rank=0, world_size=1, group=None, overlap_traces=None, force_collectives=False, pad_rows=None):
        self.tr, self.rank, self.world, self.group = tracer, rank, world_size, group
        self.force = force_collectives      # run the RCCL combine on one rank too (smoke test of the collective path)
        # a rank's share of a sharded pass is a short launch whose drain is a quarter of it: overlap consecutive
        # traces there; a whole pass on one device gains 2 % and the kernel's own duration would no longer be
        # what the events around it measure, so it keeps one trace stream
        self.overlap_traces = (world_size > 1) if overlap_traces is None else bool(overlap_traces)
        p = tracer.params
        self.ctx = [tracer.ctx, api.Context(p, tracer.gpu)]
        planes = -(-(p.nx + 2) // world_size) * world_size          # padded to a multiple of the world size
        # The private grids are this class's own, so their rows are padded to whole 64-byte lines (cbet_params.edep_zpitch).
        # With dense rows of nz + 2 = 258 doubles the pass time depends on where the grid happens to land relative to the
        # record table -- 16.9 ... 18.6 ms from one allocation to the next, reproducibly per placement (the memory channel an
        # address maps to folds address bits 7 apart: scripts/placement_sweep.py) --; with rows of 264 it does not.
        # `slabs` are views of the padded slabs with the reference's (.., ny+2, nz+2) shape.
        if pad_rows is None:
            pad_rows = int(os.environ.get("CBET_PAD_ROWS", "1"))      # 0: dense rows; 1: the next multiple of 8 doubles; > nz + 2: that pitch
        zp = int(pad_rows) if int(pad_rows) > p.nz + 2 else (-(-(p.nz + 2) // 8) * 8 if pad_rows else p.nz + 2)
        shape = (planes, p.ny + 2, zp)
        dev = tracer.device
        self.grids = [torch.zeros(shape, dtype=torch.float64, device=dev) for _ in range(2)]
        self.slab_store = [torch.zeros((planes // world_size,) + shape[1:], dtype=torch.float64, device=dev) for _ in range(2)]
        self.slabs = [s[..., : p.nz + 2] for s in self.slab_store]
        # one stream pair per buffer set: pass k+1 may start tracing while pass k is still draining
        self.s_prep = [torch.cuda.Stream(device=dev) for _ in range(2)]
        self.s_trace = [torch.cuda.Stream(device=dev) for _ in range(2)]
        if not self.overlap_traces:
            self.s_trace[1] = self.s_trace[0]
        # (one rank: the "combine" is a copy of the grid into the slab store, on the trace stream.  On a stream of its own,
        # like RCCL's, the next trace starts right behind this one -- and the pass takes 13.2-13.3 ms instead of 13.0: the copy
        # and the next pass's preparation then run beside the kernel's first wave generation.  Measured in round 5, not kept.)
        self.ev_prep = [torch.cuda.Event() for _ in range(2)]
        self.ev_consumed = [None, None]      # release(b): a reader's event the next combine into slab b waits for
        self.ev_trace = [None, None]
        self.work = [None, None]
        self.kernel_events = []
        si, sc = shard_of_rank(rank, world_size)
        self.launch_p = p.copy(beam_lo=0, beam_hi=p.nbeams, shard_index=si, shard_count=sc,
                               edep_zpitch=zp if pad_rows else 0)
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
            consumed, self.ev_consumed[b] = self.ev_consumed[b], None
            if consumed is not None:
                self.s_trace[b].wait_event(consumed)       # (behind the trace launch: only the combine waits for the reader)
            self.work[b] = reduce_scatter_grid(self.grids[b], self.slab_store[b], self.group, async_op=True, force=self.force)
            # "grid b may be cleared again": recorded AFTER the combine was enqueued -- on one rank (and with gloo) the
            # combine is a copy on this very stream, and pass k+2's grid.zero_() must not overtake it; with RCCL the
            # collective runs on the process group's stream and is waited for through its work handle
            self.ev_trace[b] = torch.cuda.Event()
            self.ev_trace[b].record()
        return b

    def wait_combined(self, b):
        """Make torch's current stream wait for the combine of buffer set b's last pass: behind it `slabs[b]` is complete (the
        RCCL collective's work handle, or the event behind the local copy)."""
        cur = torch.cuda.current_stream(self.tr.device)
        if self.work[b] is not None:
            self.work[b].wait()
        if self.ev_trace[b] is not None:
            cur.wait_event(self.ev_trace[b])

    def release(self, b):
        """A reader of `slabs[b]` on torch's current stream is done with it (enqueued so far): the next combine into that slab --
        two passes on -- waits for this point instead of relying on being later anyway."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.tr.device))
        self.ev_consumed[b] = ev

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

    def window_diagnostics(self):
        """The counters of ONE un-timed trace launch of this rank's share with cbet_params.window_stats = 1: the deposit
        windows' diagnostics (wave-steps, window misses, box-B steps, planes retired, global atomics) that the timed
        launches do not count.  Call it after counters(): it resets the contexts' counters; the last pass's slab stays."""
        tr, d = self.tr, self.tr.derived
        self.finish()
        self.counters(reset=True)
        scratch = torch.zeros_like(self.grids[0])
        with torch.cuda.stream(self.s_trace[0]):
            st = self.s_trace[0].cuda_stream
            api.trace_nodes(0, d.nindices, None, None, scratch, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r,
                            d.xconst, d.yconst, d.zconst, self.launch_p.copy(window_stats=1), self.ctx[0], st)
        torch.cuda.synchronize(tr.device)
        return self.ctx[0].counters(self.s_trace[0].cuda_stream, True)

    def warm(self):
        """Run the combine once on the (zero) buffers: RCCL builds its communicator, channels and staging buffers on
        the first collective of a kind -- set-up, like the reference's cudaMalloc in its Init phase (main.cu:131-152),
        not part of a pass.  No-op on one rank."""
        if self.world > 1 or self.force:
            for b in range(2):
                w = reduce_scatter_grid(self.grids[b], self.slab_store[b], self.group, async_op=True, force=self.force)
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

    trace_streams = 4     # the beam groups of a field pass rotate over this many streams (a launch's ramp and drain beside its neighbours')

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
    def begin_beams(self, b0, b1):
        """This rank's beams over the whole grid (their fields are complete here without any reduction) and the two
        trace streams the beam groups of a field pass alternate on."""
        tr, dev = self.tr, self.tr.device
        self.b0, self.b1 = b0, b1
        f64 = dict(dtype=torch.float64, device=dev)
        self.own_fields = torch.zeros((4, b1 - b0) + tr.grid_shape, **f64)
        self.gain_own = torch.zeros((b1 - b0,) + tr.grid_shape, **f64)
        self.gain = self.gain_own            # what a caller gets back: this rank's beams over the whole grid
        self.s_trace = [torch.cuda.Stream(device=dev) for _ in range(self.trace_streams)]
        self._launches = 0
        tr.tabulate()
        # the step records too, HERE and on this stream: left to the first launch they are built lazily on THAT launch's
        # trace stream, and the launches of the other rotating streams -- which only wait for ev_ready and see the host-side
        # version already bumped -- would read the table while k_step_table is still writing it (a fresh context: garbage)
        d = tr.derived
        api.prepare_step_records(tr.ctx, tr.params, None, None, d.xconst, d.yconst, d.zconst,
                                 torch.cuda.current_stream(dev).cuda_stream)
        self.ev_ready = torch.cuda.Event()
        self.ev_ready.record(torch.cuda.current_stream(dev))

    def begin_slab(self, pieces):
        """All beams over this rank's planes (what its gain update reads and writes): one packed array pair per PIECE
        [(lo, hi), ...] of its share of the deposit grid -- one x-slab, or two (slab_layout "paired")."""
        tr = self.tr
        nb, gs = tr.params.nbeams, tr.grid_shape
        self.pieces = [tuple(pc) for pc in pieces]
        f64 = dict(dtype=torch.float64, device=tr.device)
        self.slab_fields = [torch.zeros((4, nb, hi - lo) + gs[1:], **f64) for lo, hi in self.pieces]
        self.gain_slab = [torch.zeros((nb, hi - lo) + gs[1:], **f64) for lo, hi in self.pieces]

    def begin_slabs(self, b0, b1, x0, x1):
        self.begin_beams(b0, b1)
        self.begin_slab([(x0, x1)])

    def trace_group(self, i0, i1, use_gain, full=True, wait=()):
        """The field pass of this rank's beams [b0 + i0, b0 + i1) -- one GROUP of a pass -- on the next of the rotating
        trace streams, after the events in `wait` (the gain of these beams having arrived).  Consecutive groups overlap
        (the drain of one launch beside the head of the next), and a finished group can be sent while the next one
        traces.  Returns the event recorded behind the launch."""
        tr = self.tr
        st = self.s_trace[self._launches % len(self.s_trace)]
        self._launches += 1
        st.wait_event(self.ev_ready)
        for ev in wait:
            if ev is not None:
                st.wait_event(ev)
        with torch.cuda.stream(st):
            out = self.own_fields if full else self.own_fields[0]
            (out[:, i0:i1] if full else out[i0:i1]).zero_()
            tr.launch_cbet(out, self.gp, fields=True if full else "energy", gain=self.gain_own if use_gain else None,
                           beam_lo=self.b0 + i0, beam_hi=self.b0 + i1, grid_beam0=self.b0, grid_beams=self.b1 - self.b0)
            done = torch.cuda.Event()
            done.record(st)
        return done

    def presence_counts(self):
        """int32 [X][Y][Z]: how many of this rank's beams deposited energy at each node in the pass just traced (the
        footprint the gain update's work follows)."""
        cnt = torch.zeros(self.tr.grid_shape, dtype=torch.int32, device=self.tr.device)
        for b in range(self.b1 - self.b0):
            cnt += (self.own_fields[0, b] != 0).to(torch.int32)
        return cnt

    def slab_bytes(self):
        """Device bytes this rank's slab loop holds: the arrays of begin_beams / begin_slab and, for sparse exchanges, the
        staging buffers and segment lists (the dense exchange sends from and receives into the arrays themselves)."""
        arrays = 8 * sum(t.numel() for t in [self.own_fields, self.gain_own] + list(self.slab_fields) + list(self.gain_slab))
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
        """The whole field pass of this rank's beams as one group; the current stream waits for it."""
        if self.b1 > self.b0:
            torch.cuda.current_stream(self.tr.device).wait_event(self.trace_group(0, self.b1 - self.b0, use_gain, full))
        return self.own_fields

    def update_gain_slab(self, frozen=False, after_piece=None):
        """The gain update of all beams over this rank's pieces; a second piece runs on a side stream beside the first (two
        small launches one after the other would each pay their own ramp and drain).  after_piece(k, event): the pieces
        then run ONE AFTER THE OTHER on the current stream, `event` recorded behind piece k -- the caller sends piece k's
        gain back while piece k + 1 updates.  (All launches are enqueued before the first call-back: enqueueing a piece's
        messages takes the host longer than the piece's update takes the device.)"""
        self.change.zero_()
        cur = torch.cuda.current_stream(self.tr.device)
        gp = _frozen(self.gp, frozen)
        if after_piece is not None:
            done = []
            for (lo, hi), fields, gain in zip(self.pieces, self.slab_fields, self.gain_slab):
                if hi > lo:
                    api.gain_field_packed(fields, None, gain, gain, self.change, lo, hi, self.tr.params, gp, self.tr.ctx, cur.cuda_stream)
                done.append(torch.cuda.Event())
                done[-1].record(cur)
            for k, ev in enumerate(done):
                after_piece(k, ev)
            return self.change
        side_done = []
        for k, ((lo, hi), fields, gain) in enumerate(zip(self.pieces, self.slab_fields, self.gain_slab)):
            if hi <= lo:
                continue
            if k == 0:
                api.gain_field_packed(fields, None, gain, gain, self.change, lo, hi, self.tr.params, gp, self.tr.ctx, cur.cuda_stream)
            else:
                if not hasattr(self, "s_side"):
                    self.s_side = torch.cuda.Stream(device=self.tr.device)
                ready = torch.cuda.Event()
                ready.record(cur)                      # (behind the zeroing of `change` and the arrival fence)
                self.s_side.wait_event(ready)
                api.gain_field_packed(fields, None, gain, gain, self.change, lo, hi, self.tr.params, gp, self.tr.ctx, self.s_side.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(self.s_side)
                side_done.append(ev)
        for ev in side_done:
            cur.wait_event(ev)
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


def balanced_slabs(weights, world_size, widest=None):
    """Contiguous plane ranges [(lo, hi)] per rank whose summed `weights` (one per plane of the haloed grid) are as equal as
    whole planes allow; every rank gets at least one plane while there are enough.  widest: no slab wider than this many
    planes (a per-plane constant is added to the weights until that holds: the cut moves towards equal plane counts) --
    the time of a grouped send/recv is set by its largest message, i.e. by the widest slab, so work balance is bought with
    link time.  The same deterministic rule on every rank (the weights come out of an all-reduce of integers)."""
    w = np.asarray(weights, dtype=np.float64)
    X = len(w)
    if world_size >= X or not np.isfinite(w).all() or w.sum() <= 0:
        return _parts(X, world_size)

    def cut(v):
        cum = np.concatenate([[0.0], np.cumsum(v)])
        cuts = [0]
        for r in range(1, world_size):
            target = cum[-1] * r / world_size
            x = int(np.searchsorted(cum, target))
            if x > 0 and abs(cum[x - 1] - target) <= abs(cum[min(x, X)] - target):
                x -= 1
            cuts.append(min(max(x, cuts[-1] + 1), X - (world_size - r)))
        cuts.append(X)
        return [(cuts[r], cuts[r + 1]) for r in range(world_size)]

    out = cut(w)
    if widest is not None:
        widest = max(int(widest), -(-X // world_size))
        lam, mean = 0.0, w.mean()
        for _ in range(40):
            if max(hi - lo for lo, hi in out) <= widest:
                break
            lam = mean * 0.05 if lam == 0.0 else lam * 1.5
            out = cut(w + lam)
        else:
            out = _parts(X, world_size)
    return out


def slab_pieces(layout, nx_halo, world_size, weights=None):
    """Which planes of the deposit grid each rank's gain update owns: [[(lo, hi), ...] per rank].
    "equal"  : one x-slab per rank, equal plane counts (the beams cross at the centre: the central ranks' update takes
               twice the outer ranks');
    "paired" : the grid is cut into 2 W equal blocks and rank r owns block r AND block W + r -- one from the left half
               counted from the edge, one from the right half counted from the centre, i.e. a light and a heavy one: the work
               evens out (modelled cost spread 1.53 -> 1.13 at 256^3 / 60 beams / 8 ranks) while every message keeps the
               same size, so no link carries more than another (twice the messages, half as long);
    a number > 1 : one slab per rank cut by the modelled gain-update cost `weights`, none wider than that multiple of the
               equal share (balanced_slabs) -- balances better and lengthens every grouped call by the widest slab;
    "halves" : the equal slab as TWO pieces (lower and upper half of its planes): the loop updates them one after the other
               and sends the first half's gain back while the second half updates (never priced on real links)."""
    if layout == "halves":
        out = []
        for lo, hi in _parts(nx_halo, world_size):
            mid = lo + (hi - lo + 1) // 2
            out.append([(lo, mid), (mid, hi)])          # (always two, an empty one included: the ranks exchange piece by piece)
        return out
    if layout == "paired" and world_size > 1:
        blocks = _parts(nx_halo, 2 * world_size)
        return [[b for b in (blocks[r], blocks[world_size + r]) if b[1] > b[0]] or [(0, 0)] for r in range(world_size)]
    if isinstance(layout, (int, float)) and not isinstance(layout, bool) and float(layout) > 1.0 and world_size > 1 and weights is not None:
        return [[pc] for pc in balanced_slabs(weights, world_size, widest=float(layout) * nx_halo / world_size)]
    return [[pc] for pc in _parts(nx_halo, world_size)]


def gain_update_weights(counts):
    """Per-plane cost of the gain update from the number of beams present at each node (`counts`, integer [X][Y][Z], summed
    over ranks).  The pair-once kernel works in runs of 16 cells along z whose cells advance in lockstep, so a run costs what
    its most crowded cell costs, and that grows with the SQUARE of the beams there (pairs): timed on 8-plane slabs of the
    256^3 / 60-beam grid, a plane costs 0.048 ms where the runs' maxima average m^2 = 154 and 0.125 ms where they average
    284 -- 2.6 x for 1.2 x the beams per node -- and ms per plane = 6.2e-4 (mean m^2 - 78) fits to 12 %
    (scripts/gain_plane_cost.py, profiles/r4/gain_plane_cost.log).  Returns a float64 numpy array [X] (relative weights)."""
    n = counts.to(torch.float64)
    X, Y, Z = n.shape
    pad = (-Z) % 16
    m = torch.nn.functional.pad(n, (0, pad)).view(X, Y, (Z + pad) // 16, 16).max(-1).values
    return (m * m - 78.0).clamp_(min=8.0).sum((1, 2)).cpu().numpy()


class _SlabExchanger:
    """The two all-to-all exchanges of the slab-owned CBET loop over point-to-point xGMI links (RCCL send/recv; gloo in
    the CPU tests), one message per (beam, peer, component) and NO staging: the part of a beam over an x-slab is
    contiguous both in the sender's whole-grid array and in the receiver's slab array, so messages go from and into the
    arrays themselves.

    * All W - 1 peers at once: the sends of my i-th beam to every slab owner and the receives of every peer's i-th beam
      are ONE grouped send/recv (batch_isend_irecv = ncclGroupStart ... End), so all seven links of a rank carry a
      message at the same time.  Chunks are beams (and components), never peers.
    * Stream-ordered: everything runs on a communication stream that waits for the producer's event (the trace of the
      beam's group; the gain update) and hands an event to the consumer (the gain update; the next trace of the group) --
      no host synchronisation.  A group's fields travel while the next group traces; a group's gain comes back while
      the previous groups already trace the next pass.
    * gloo has no device path: device tensors are staged through the host message by message (tests only).
    Ranks without beams or planes simply post nothing; every rank walks the beam indices in the same order, so the
    sends and receives of a pair match in order."""

    def __init__(self, device, group, rank, world_size, beams, force_collectives=False, emulate=None, two_channels=False):
        import torch.distributed as dist
        self.group, self.device, self.rank, self.world, self.beams = group, torch.device(device), rank, world_size, beams
        self.cuda = self.device.type == "cuda"
        self.dist_on = dist.is_available() and dist.is_initialized()
        # emulate(exchanger, sends, recvs): a stand-in for the transport of one grouped send/recv, run on the communication
        # stream exactly where RCCL's would be (scripts/cbet_rank_share.py: one rank's schedule on one GPU, the peers'
        # data supplied and the link time priced) -- everything else of the schedule is the product's
        self.emulate = emulate
        self.nccl = self.cuda and ((self.dist_on and dist.get_backend(group) == "nccl") or emulate is not None)
        self.force = force_collectives      # one rank: the same send/recv machinery as a self-exchange (RCCL smoke test)
        self.stream = torch.cuda.Stream(device=self.device) if self.nccl else None
        # two_channels: exchange 2 (the gain's way back) gets a communicator and a stream of its own, so that a pass's fields
        # do not queue behind the previous pass's gain on one in-order channel (every rank creates the second group here, in
        # the same place of its program: new_group is collective)
        self.group2, self.stream2 = group, self.stream
        self.two_channels = False
        if two_channels and emulate is not None:
            self.stream2, self.two_channels = (torch.cuda.Stream(device=self.device) if self.nccl else None), True
        elif two_channels and self.dist_on and (world_size > 1 or force_collectives):
            self.two_channels = True
            ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
            self.group2 = dist.new_group(ranks=ranks, backend=dist.get_backend(group))
            self.stream2 = torch.cuda.Stream(device=self.device) if self.nccl else None
        self.solo = world_size == 1
        self.peers = [rank] if (self.solo and self.force) else [r for r in range(world_size) if r != rank]
        self.slabs = None
        self.plan = None
        self.send_buf = self.recv_buf = None    # sparse exchanges only
        self.chunks = self.messages = self.bytes_sent = 0

    def set_slabs(self, pieces):
        """pieces[r] = the plane ranges [(lo, hi), ...] rank r's gain update owns (slab_pieces)."""
        self.slabs = pieces

    def staging_bytes(self):
        if self.send_buf is None or self.send_buf.device != self.device:
            return 0
        return 8 * (self.send_buf.numel() + self.recv_buf.numel())

    def _global(self, r):
        import torch.distributed as dist
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _enter(self, after, back=False):
        """Order what follows behind the events in `after`: on the communication stream (RCCL) -- the second channel's for
        the gain's way back (`back`) -- or the current one."""
        cs = self.stream2 if back else self.stream
        target = cs if self.nccl else (torch.cuda.current_stream(self.device) if self.cuda else None)
        if target is not None:
            for ev in after:
                if ev is not None:
                    target.wait_event(ev)
        return torch.cuda.stream(cs) if self.nccl else _NullContext()

    def _leave(self, back=False):
        """An event behind everything issued so far on that channel (None without a device)."""
        if not self.cuda:
            return None
        ev = torch.cuda.Event()
        ev.record((self.stream2 if back else self.stream) if self.nccl else torch.cuda.current_stream(self.device))
        return ev

    def fence(self):
        return self._leave()

    def _batch(self, sends, recvs, back=False):
        """One grouped send/recv: `sends` / `recvs` are (tensor view, peer) lists of contiguous views."""
        import torch.distributed as dist
        group = self.group2 if back else self.group
        if not sends and not recvs:
            return
        if self.emulate is not None:
            self.emulate(self, sends, recvs)
            self.bytes_sent += 8 * sum(t.numel() for t, _ in sends)
            self.chunks += 1
            self.messages += len(sends) + len(recvs)
            return
        ops, late = [], []
        for t, peer in sends:
            if self.cuda and not self.nccl:
                t = t.cpu()                         # gloo: through the host (synchronises the current stream)
            ops.append(dist.P2POp(dist.isend, t, self._global(peer), group))
            self.bytes_sent += 8 * t.numel()
        for t, peer in recvs:
            if self.cuda and not self.nccl:
                h = torch.empty(t.shape, dtype=t.dtype, device="cpu")
                late.append((t, h))
                t = h
            ops.append(dist.P2POp(dist.irecv, t, self._global(peer), group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()     # RCCL: the communication stream waits (no host block); gloo: the host waits
        for t, h in late:
            t.copy_(h)
        self.chunks += 1
        self.messages += len(ops)

    def fields_out(self, own, slab, i0, i1, comps, after=()):
        """Exchange 1 for the beams with index [i0, i1) of every rank: my beams' fields over rank s's pieces -> rank s, rank
        q's beams over my pieces <- rank q, component by component of `comps`.  own: [4][my beams][X][Y][Z]; slab: one
        [4][all beams][piece planes][Y][Z] per piece of mine."""
        rank, beams, pieces = self.rank, self.beams, self.slabs
        b0, b1 = beams[rank]
        with self._enter(after):
            for i in range(i0, i1):
                sends, recvs = [], []
                mine = i < b1 - b0
                for s_ in self.peers:
                    if mine:
                        sends += [(own[c, i, lo:hi], s_) for lo, hi in pieces[s_] if hi > lo for c in comps]
                    q0, q1 = beams[s_]
                    if i < q1 - q0:
                        recvs += [(slab[k][c, q0 + i], s_) for k, (lo, hi) in enumerate(pieces[rank]) if hi > lo for c in comps]
                if mine and not (self.solo and self.force):
                    for k, (lo, hi) in enumerate(pieces[rank]):
                        for c in comps:
                            if hi > lo:
                                slab[k][c, b0 + i].copy_(own[c, i, lo:hi])       # the own part never travels
                self._batch(sends, recvs)
        return self._leave()

    def gain_back(self, gain_slab, gain_own, i0, i1, after=(), only_piece=None):
        """Exchange 2 for the beams with index [i0, i1): the new gain of rank q's beams over my pieces -> rank q, my beams'
        gain over rank s's pieces <- rank s.  gain_slab: one [all beams][piece planes][Y][Z] per piece of mine; gain_own:
        [my beams][X][Y][Z].  only_piece = k: the k-th piece of EVERY rank alone (the loop sends a half slab's gain while the
        other half still updates; every rank has the same number of pieces then).  Runs on the second channel when there is
        one.  Returns the event behind it: the next pass's trace of these beams waits for it."""
        rank, beams, pieces = self.rank, self.beams, self.slabs
        b0, b1 = beams[rank]

        def want(k):
            return only_piece is None or k == only_piece
        with self._enter(after, back=True):
            for i in range(i0, i1):
                sends, recvs = [], []
                mine = i < b1 - b0
                for q in self.peers:
                    q0, q1 = beams[q]
                    if i < q1 - q0:
                        sends += [(gain_slab[k][q0 + i], q) for k, (lo, hi) in enumerate(pieces[rank]) if hi > lo and want(k)]
                    if mine:
                        recvs += [(gain_own[i, lo:hi], q) for k, (lo, hi) in enumerate(pieces[q]) if hi > lo and want(k)]
                if mine and not (self.solo and self.force):
                    for k, (lo, hi) in enumerate(pieces[rank]):
                        if hi > lo and want(k):
                            gain_own[i, lo:hi].copy_(gain_slab[k][b0 + i])
                self._batch(sends, recvs, back=True)
        return self._leave(back=True)

    # ---- the sparse form: only the 64-byte z-runs inside the beams' footprints move (SegmentPlan) --------------------
    def use_plan(self, plan):
        """Sparse exchanges: staging for the runs of ALL peers of one component at once, out and in."""
        self.plan = plan
        stage_dev = self.device if (self.nccl or not self.cuda) else torch.device("cpu")
        peers = self.peers
        n_out = 8 * max(sum(plan.own_side[s].shape[0] for s in peers), sum(plan.slab_side[q].shape[0] for q in peers))
        self.send_buf = torch.empty(n_out, dtype=torch.float64, device=stage_dev)
        self.recv_buf = torch.empty(n_out, dtype=torch.float64, device=stage_dev)

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

    def run_sparse(self, src, send_index, dst, recv_index, to_slabs, ncomp=0, after=()):
        """One exchange moving only the z-runs of the plan: to_slabs = exchange 1 (pack from my whole-grid array, unpack
        into my slab array), else exchange 2.  ncomp > 0: the arrays carry that many leading components.  Per component:
        the runs of ALL peers are packed into consecutive stretches of the send staging buffer, travel in one grouped
        send/recv, and are unpacked from the receive staging buffer.  Stream-ordered like the dense form."""
        import torch.distributed as dist
        plan, rank = self.plan, self.rank
        hy, hz = plan.Y, plan.Z
        out_lists, in_lists = (plan.own_side, plan.slab_side) if to_slabs else (plan.slab_side, plan.own_side)
        out_stride, in_stride = (plan.own_stride, plan.slab_stride) if to_slabs else (plan.slab_stride, plan.own_stride)
        dev_stage = self.send_buf.device == src.device
        with self._enter(after):
            if not (self.solo and self.force):
                dst[recv_index(rank)] = src[send_index(rank)]       # the own part: a dense local copy
            for c in range(max(1, ncomp)):
                s_arr = src[c] if ncomp else src
                d_arr = dst[c] if ncomp else dst
                ops, off_out, off_in, unpack = [], 0, 0, []
                for peer in self.peers:
                    seg_out, seg_in = out_lists[peer], in_lists[peer]
                    n_out, n_in = seg_out.shape[0], seg_in.shape[0]
                    if n_out:
                        sb = self.send_buf[off_out: off_out + 8 * n_out]
                        if dev_stage:
                            self._pack(s_arr, out_stride, hy, hz, seg_out, sb)
                        else:                   # gloo with device arrays: pack on the device, stage through the host
                            tmp = torch.empty(8 * n_out, dtype=torch.float64, device=src.device)
                            self._pack(s_arr, out_stride, hy, hz, seg_out, tmp)
                            sb.copy_(tmp)
                        ops.append(dist.P2POp(dist.isend, sb, self._global(peer), self.group))
                        self.bytes_sent += 64 * n_out
                        off_out += 8 * n_out
                    if n_in:
                        rb = self.recv_buf[off_in: off_in + 8 * n_in]
                        ops.append(dist.P2POp(dist.irecv, rb, self._global(peer), self.group))
                        unpack.append((seg_in, rb))
                        off_in += 8 * n_in
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                    self.chunks += 1
                    self.messages += len(ops)
                for seg_in, rb in unpack:
                    self._unpack(d_arr, in_stride, hy, hz, seg_in, rb if dev_stage else rb.to(dst.device))
        return self._leave()


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


def cbet_fixed_point_slabs(engine, gain_params, nbeams, nx_halo, rank=0, world_size=1, group=None, sparse=False,
                           trace_groups=4, slab_layout="equal", two_channels=False):
    """The CBET fixed-point iteration with storage, exchange and schedule sized for 8 ranks on point-to-point xGMI (SURVEY
    8(f) f1; parity unpinned; same passes and same result as cbet_fixed_point).

    Rank r traces WHOLE beams [b_r0, b_r1) -- their four fields are complete on r without any reduction -- and
    owns some planes of the deposit grid for the gain update (its PIECES, slab_pieces).  It STORES only
        its own beams over the whole grid : own_fields [4][nb_r][X][Y][Z], gain_own [nb_r][X][Y][Z]
        all beams over its own pieces     : per piece slab_fields [4][nb][x_p][Y][Z], gain_slab [nb][x_p][Y][Z]
    (cbet_cbet_slab_workspace_bytes_parts).  slab_layout says which planes.  "equal" (default): one x-slab per rank, equal
    plane counts -- the beams cross at the centre, a central plane's update costs twice an outer one's, and the central
    ranks' update takes 2.4 ms against the outer ranks' 1.2 (256^3 / 60 beams / 8 ranks) while everybody waits for it.
    "paired": every rank owns one block of the grid's left half counted from the edge and one of its right half counted
    from the centre -- a light and a heavy one, messages of equal size (twice as many, half as long): measured 1.9-2.25 ms
    on every rank (two 16-plane launches each pay their ramp and drain), i.e. 0.15 ms off the slowest.  A number > 1: one
    slab per rank cut by the modelled cost (gain_update_weights from the beams counted per node after the first field
    pass, all-reduced once), none wider than that multiple of the equal share: 1.7-2.0 ms everywhere when unlimited, but a
    grouped send/recv lasts as long as its largest message, which goes to the widest slab (1.6 x), and at the 64 GB/s the
    links are priced with that costs more than the balance saves.  Neither moves the iteration outside the noise of the
    one-GPU emulation (scripts/cbet_rank_share.py, profiles/r4/cbet_rank_share.log); the simplest stays the default.
    "halves": the equal slab updated as its lower and its upper half one after the other, the lower half's gain on its way
    back while the upper half updates.  two_channels=True: the gain's way back gets its own communicator and stream, so a
    pass's fields never queue behind gain messages.  Both are schedule changes only (same arithmetic on the same cells;
    equality tests at 2 and 3 ranks) and neither has been priced: that needs real links, the one-GPU emulation shares one
    copy engine between the "channels".

    One pass, pipelined over beam GROUPS (trace_groups of them; engine.trace_group rotates them over its trace streams):
        trace group g  ->  exchange 1 of group g's beams (while the later groups trace): my beams' fields over slab s to
        rank s, all peers of a beam in one grouped send/recv  ->  [all groups in]  gain update of ALL beams on my slab
        ->  exchange 2, group by group in the order the next pass traces them: the gain of rank q's beams over my
        slab back to q  ->  the next pass's trace of group g starts as soon as ITS gain is in.
    The two scalars of the convergence measure are all-reduced and rank 0's copy decides, while exchange 2 is already
    in flight.  At 256^3 / 60 beams / 8 ranks a rank sends 3.6 GB + 0.9 GB in the direction-building first pass and
    0.9 GB + 0.9 GB in every later one (energy field only).  sparse=True: the un-pipelined exchange of only the z-runs
    inside the beams' footprints (SegmentPlan; exact; does not pay for this physics, see profiles/r3/cbet_rank_share.log).
    `engine`: begin_beams(b0, b1); trace_group(i0, i1, use_gain, full, wait) -> event or None; presence_counts() ->
    integer [X][Y][Z]; begin_slab(pieces); attributes own_fields, gain_own, slab_fields, gain_slab (lists, one per piece);
    update_gain_slab(frozen[, after_piece]) -> tensor {sum |dK|, sum |K|} over the slab; deposit_beams() -> beam_gain.  The
    deposition grid is left un-reduced (allreduce_grid / reduce_scatter_grid)."""
    import torch.distributed as dist
    beams = _parts(nbeams, world_size)
    b0, b1 = beams[rank]
    nbr = b1 - b0
    imax = max(q1 - q0 for q0, q1 in beams)
    groups = [g for g in _parts(imax, max(1, min(trace_groups, imax))) if g[1] > g[0]]     # beam-INDEX ranges, the same on every rank
    force = getattr(engine, "force_collectives", False)   # one rank, but every collective really runs (RCCL smoke test)
    engine.begin_beams(b0, b1)
    xch = _SlabExchanger(engine.own_fields.device, group, rank, world_size, beams, force_collectives=force,
                         emulate=getattr(engine, "emulate_transport", None), two_channels=two_channels)
    engine.exchanger = xch
    on_device = engine.own_fields.is_cuda
    cur = (lambda: torch.cuda.current_stream(engine.own_fields.device)) if on_device else None

    def wait_here(events):
        if on_device:
            for ev in events:
                if ev is not None:
                    cur().wait_event(ev)

    def mark():
        if not on_device:
            return None
        ev = torch.cuda.Event()
        ev.record(cur())
        return ev

    def all_reduce_host_staged(t):
        if world_size > 1 or force:
            if t.is_cuda and dist.get_backend(group) != "nccl":
                host = t.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                return host
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t

    # sparse = True: the exchanges move only the 64-byte z-runs a beam's rays can ever touch instead of dense sub-arrays.
    # Exact, but it does not pay for this physics: 73 % of the nodes are inside a beam's footprint.  (The footprint pass
    # uses the field arrays as scratch and resets the counters: before the first field pass.)
    support = engine.support_mask() if (sparse and hasattr(engine, "support_mask") and (world_size > 1 or force)) else None
    slabs, pieces, plan = None, None, None
    allocated = None    # event behind the first-pass allocation (zero-fill) of the slab arrays
    G = len(groups)
    gain_ev = [None] * G
    owed = []           # groups whose gain of the previous pass has not been sent back yet (it goes out between this pass's traces)
    updated = None
    # ONE in-order channel carries both exchanges (RCCL serialises the calls of a communicator), so the order they are
    # enqueued in is the schedule: behind the update the gain of ALL groups goes back, group by group in the order the next
    # pass traces them and before the host has even looked at the convergence scalars; group k of the next pass starts
    # tracing when ITS gain is in (k + 1 calls into the exchange) and the groups overlap on the rotating trace streams;
    # a group's fields are enqueued right behind its trace and travel while the later groups trace.  (Measured with the
    # kernel trace of scripts/cbet_rank_share.py: sending a group's gain just ahead of its own trace, interleaved with
    # the fields of earlier groups, chains trace -> fields -> gain -> trace through the one channel and stretches the
    # trace phase from 4.5 to 9 ms.)  HEAD = groups whose gain goes back before the scalars are looked at, LAG = groups
    # between a trace and the enqueueing of its fields.
    HEAD, LAG = G, 0

    def send_gain(k):
        i0, i1 = groups[k]
        return xch.gain_back(engine.gain_slab, engine.gain_own, i0, i1, after=(updated,))

    rep = {"passes": 0, "converged": False, "change": float("inf")}
    for it in range(gain_params.max_passes):
        full = it < gain_params.direction_passes
        comps = range(4) if full else range(1)    # after the direction-building passes only the energy field moves
        traced = [None] * G
        for k, (i0, i1) in enumerate(groups):
            if k in owed:
                gain_ev[k] = send_gain(k)
                owed.remove(k)
            j0, j1 = min(i0, nbr), min(i1, nbr)
            if j1 > j0:
                traced[k] = engine.trace_group(j0, j1, it > 0, full, wait=(gain_ev[k],))
            if slabs is not None and plan is None and k >= LAG:
                xch.fields_out(engine.own_fields, engine.slab_fields, *groups[k - LAG], comps, after=(traced[k - LAG], updated))
        if slabs is None:
            # first pass: the beams' footprints are known now -- cut the slabs, allocate them, then send everything
            wait_here(traced)
            layout = "equal" if support is not None else slab_layout     # (the sparse plan is written for one slab per rank)
            weights = None
            if isinstance(layout, (int, float)) and not isinstance(layout, bool) and float(layout) > 1.0 and world_size > 1:
                counts = all_reduce_host_staged(engine.presence_counts())
                weights = gain_update_weights(counts)
                del counts
            pieces = slab_pieces(layout, nx_halo, world_size, weights)
            slabs = [pcs[0] for pcs in pieces]                            # (what the sparse plan indexes by)
            engine.begin_slab(pieces[rank])
            xch.set_slabs(pieces)
            if support is not None:
                plan = SegmentPlan(support, beams, slabs, rank, world_size, group, engine.slab_fields[0].device)
                support = None
                xch.use_plan(plan)
            # begin_slab / use_plan zero-fill their arrays on THIS stream, which has just been made to wait for every trace
            # group; the exchanges below run on the communication stream and wait for their own group's trace only -- without
            # this event group 0's fields would land in slab_fields before the (late) fill and be zeroed by it
            allocated = mark()
            first = 0
        else:
            first = G - LAG
        if plan is not None:
            xch.run_sparse(engine.own_fields, lambda s_: (slice(0, len(comps)), slice(None), slice(*slabs[s_])),
                           engine.slab_fields[0], lambda q: (slice(0, len(comps)), slice(*beams[q])), True,
                           ncomp=len(comps), after=list(traced) + [allocated])
        else:
            for k in range(first, G):
                # (`updated`: a rank without a beam in this group has no trace to wait for, and with two channels nothing
                # else keeps the peers' fields out of slab_fields while the previous update still reads them)
                xch.fields_out(engine.own_fields, engine.slab_fields, *groups[k], comps, after=(traced[k], allocated, updated))
        allocated = None
        wait_here([xch.fence()])                     # every beam's fields over my slab are in
        if xch.two_channels:
            wait_here(gain_ev)                       # ... and the previous gain has left gain_slab (one channel: in order)
        split = slab_layout == "halves" and plan is None

        def piece_done(kp, ev):
            # the gain of piece kp is final behind `ev`: all groups' share of it goes back, while the next piece updates
            for k in range(G):
                gain_ev[k] = xch.gain_back(engine.gain_slab, engine.gain_own, *groups[k], after=(ev,), only_piece=kp)
        ch = engine.update_gain_slab(not full, after_piece=piece_done) if split else engine.update_gain_slab(not full)
        updated = mark()
        # the gain of rank q's beams over my slab -> rank q; my beams' gain over slab s <- rank s.  It is due whatever the
        # convergence scalars say (the deposition pass needs the new gain too): the head goes out now
        if plan is not None:
            ev = xch.run_sparse(engine.gain_slab[0], lambda q: (slice(*beams[q]),),
                                engine.gain_own, lambda s_: (slice(None), slice(*slabs[s_])), False, after=(updated,))
            gain_ev, owed = [ev] * G, []
        elif split:
            owed = []                                # (went back piece by piece, see piece_done)
        else:
            for k in range(HEAD):
                gain_ev[k] = send_gain(k)
            owed = list(range(HEAD, G))
        ch = _agree(all_reduce_host_staged(ch), group, world_size)
        rep["passes"] = it + 1
        rep["change"] = float(ch[0] / ch[1]) if float(ch[1]) > 0 else 0.0
        if rep["change"] < gain_params.tolerance:
            rep["converged"] = True
            break
    for k in list(owed):                              # no further pass: the rest of the gain goes back now
        gain_ev[k] = send_gain(k)
    wait_here(gain_ev)
    beam_gain = engine.deposit_beams()
    if world_size > 1 or force:
        allreduce_grid(beam_gain, group, force)
    bg = beam_gain.cpu().numpy().copy()
    rep["beam_gain"] = bg
    rep["imbalance"] = float(abs(bg.sum()) / np.abs(bg).sum()) if np.abs(bg).sum() > 0 else 0.0
    rep["slabs"] = pieces
    rep["groups"] = groups
    return rep


def traced_pass(tracer, edep, rank=0, world_size=1, group=None, **launch_kw):
    """One full pass over the beams on `world_size` ranks: zero, trace this rank's share, combine."""
    edep.zero_()
    si, sc = shard_of_rank(rank, world_size)
    tracer.launch(edep, shard_index=si, shard_count=sc, **launch_kw)
    return allreduce_grid(edep, group)
