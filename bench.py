#!/usr/bin/env python3
"""Headline benchmark: ray-steps/sec of the OMEGA 60-beam sweep (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full pass of the hot path over the workload: zero the deposition grid, tabulate the
plasma node tables and the per-node step records, trace this rank's share of the ray bundles of all 60
beams, combine the grids (RCCL reduce-scatter into x-slabs) when N > 1.  The K passes of a run are
independent, so two are kept in flight (tracer.SweepPipeline): the next pass's tables are prepared
beside the drain of the current trace and its combine runs beside the next trace; the timed region
brackets all K passes, combines included.  Inputs are resident in HBM before the timed region.  The
workload is fixed as N grows (60 beams sharded N ways) -> "scaling": "strong".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_RAY_STEP = 128      # 8 fp64 gathers + 8 fp64 atomic-add payloads (context only, see roofline.note)
HBM_PEAK = 8.0e12             # MI355X_MICROARCH.md: HBM3E spec B/s
HBM_COPY_RATE = 6.29e12       # ... and the copy rate the same guide measures as achievable
SIMDS, CLOCK_HZ = 1024, 2.4e9 # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, max clock
VALU_ISSUE_PEAK = SIMDS * CLOCK_HZ / 4.0     # one wave-instruction per SIMD per 4 cycles ("vector-instruction ISSUE cost")
ATOMIC_PEAK = 1.3e12          # MI355X_MICROARCH.md "Global float atomics": bytes/s of 64-B memory-side atomic requests
LDS_CYCLE_PEAK = 256 * CLOCK_HZ  # one LDS array per CU: CU-cycles per second the LDS could be busy


COMBINE_NOTE = ("two passes in flight per rank on three HIP streams (prepare tables | trace | RCCL reduce-scatter of the "
                "(n+2)^3 fp64 grid into x-slabs), tracer.SweepPipeline")


def cpu_baseline(n, r, ne, te, bn):
    """The serial CPU ray loop (oracle/cbet_oracle.c, 1 thread) on a bounded sample of the same
    workload: the first beams of the n^3 sweep (~1e8 ray-steps, 10-30 s)."""
    from oracle import cbet_oracle as O
    cfg = O.default_config(n)
    nb = 3 if n >= 200 else 60
    t0 = time.perf_counter()
    _, steps = O.trace(cfg, bn, r, ne, te, beam_lo=0, beam_hi=nb, nthreads=1)
    dt = time.perf_counter() - t0
    ncores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    t0 = time.perf_counter()
    _, steps_mt = O.trace(cfg, bn, r, ne, te, beam_lo=0, beam_hi=nb, nthreads=ncores)
    dt_mt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "ray-steps/s", "cores": 1, "kind": "port",
            "sample": "beams 0-%d of the %d^3 60-beam s83177 sweep: %d ray-steps in %.1f s, "
                      "serial ray loop, 1 thread, gcc -O2 -ffp-contract=off" % (nb - 1, n, steps, dt),
            # BASELINE.md section 3's second figure: the same sample on every host core (OpenMP over rays)
            "all_cores_value": steps_mt / dt_mt, "all_cores": ncores,
            "all_cores_note": "threads capped at 16, the CPU share of a 1-GPU job on this pool"}


def cbet_leg(api, tr, edep, n):
    """The CBET iteration on the same workload, reported BESIDE the headline and never
    part of `value`: the reference has no CBET code, so this stage is parity-unpinned (checked against
    the CPU restatement of its model and by energy conservation, tests/test_gpu_cbet.py)."""
    import numpy as np
    import torch
    gp = api.default_gain_params()
    try:
        ws = torch.empty(api.cbet_workspace_bytes(tr.params) // 8, dtype=torch.float64, device=edep.device)
    except RuntimeError as exc:   # not enough free HBM next to whatever else the process holds
        return {"skipped": "workspace allocation failed: %s" % str(exc).splitlines()[0]}
    stream = torch.cuda.current_stream().cuda_stream
    args = (tr.d_te, tr.d_r, tr.d_ne, edep, tr.d_bbeam_norm, tr.d_beam_norm, tr.d_pow_r, tr.d_phase_r, tr.params, gp)
    edep.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = api.cbet_solve(*args, workspace=ws, ctx=tr.ctx, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bg = np.array(rep.beam_gain[:tr.params.nbeams])
    absorbed = float(edep.sum().item())
    # one steady-state iteration again on the solve's own arrays, each kernel timed with HIP events: the energy-field pass
    # of all beams with the converged gain, then the gain update with frozen directions -- priced with the counts of the
    # committed rocprofv3 --pmc passes of the same two kernels (profiles/r*/cbet/traffic.json)
    nb, hs = tr.params.nbeams, int(np.prod(tr.grid_shape))
    fields = ws[: 4 * nb * hs].view((4, nb) + tr.grid_shape)
    gain = ws[4 * nb * hs: 5 * nb * hs].view((nb,) + tr.grid_shape)
    change = torch.zeros(2, dtype=torch.float64, device=edep.device)
    t_field, t_gain = [], []
    for _ in range(3):
        fields[0].zero_()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        tr.launch_cbet(fields[0], gp, fields="energy", gain=gain)
        e[1].record()
        e[2].record()
        tr.gain_field(fields, gain, gp, change, pair_once=True, frozen=True)
        e[3].record()
        torch.cuda.synchronize()
        t_field.append(e[0].elapsed_time(e[1]) * 1e-3)
        t_gain.append(e[2].elapsed_time(e[3]) * 1e-3)
    t_field, t_gain = sum(t_field[1:]) / 2, sum(t_gain[1:]) / 2
    prof = {}
    pdir = os.path.join(ROOT, "profiles")
    for rnd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        path = os.path.join(pdir, rnd, "cbet", "traffic.json")
        if os.path.exists(path) and n == 256 and nb == 60:
            for ent in json.load(open(path)).get("entries", []):
                prof["gain" if "gain" in ent["kernel"] else "field"] = ent
    alg_gain = 5.0 * nb * hs * 8     # the whole workspace once: four field components and the gain

    def priced(ent, seconds, extra):
        out = {"kernel_ms": 1e3 * seconds, **extra}
        if ent is not None:
            out.update({"bound": "hbm", "achieved": ent["hbm_bytes_per_launch"] / seconds / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": ent["hbm_bytes_per_launch"] / seconds / HBM_PEAK, "traffic": ent["hbm_bytes_per_launch"],
                        "frac_of_measured_copy_rate": ent["hbm_bytes_per_launch"] / seconds / HBM_COPY_RATE,
                        "valu_issue_frac": ent["SQ_INSTS_VALU_per_launch"] / seconds / VALU_ISSUE_PEAK,
                        "wave_cycles_waiting_frac": ent["SQ_WAIT_ANY_per_launch"] / ent["SQ_WAVE_CYCLES_per_launch"],
                        "traffic_source": ent["source"]})
        return out

    return {"parity": "unpinned (no reference CBET code; model of DESIGN.md section 9)",
            "iteration": {
                "gain_kernel": priced(prof.get("gain"), t_gain, {
                    "kernel": "k_gain_field_sym", "algorithmic_bytes": alg_gain, "algorithmic_GBps": alg_gain / t_gain / 1e9,
                    "note": "every entry is read once, the cell's beams are staged in LDS, runs start on the arrays' 128-byte lines "
                            "(DESIGN.md section 9): measured traffic = the workspace (41.2 GB) to 1 % in a frozen-direction call; "
                            "frac = (2 x FETCH_SIZE + WRITE_SIZE) / kernel time / 8 TB/s, frac_of_measured_copy_rate against the "
                            "6.29 TB/s the guide measures as achievable.  No ceiling is near: HBM is the nearest, vector issue "
                            "(valu_issue_frac) next, and more than half of the wave cycles wait (wave_cycles_waiting_frac): the "
                            "load / pair / store phases of a run overlap only with other waves' at 12 waves per CU"}),
                "energy_field_pass": priced(prof.get("field"), t_field, {"kernel": "k_trace_window<16,false,2>"}),
                "ms": 1e3 * (t_field + t_gain)},
            "workload": "omega60_%dcube_s83177_absorption + CBET fixed-point iteration" % n,
            "passes": rep.passes, "converged": bool(rep.converged), "gain_change": rep.change,
            "energy_imbalance": rep.imbalance, "seconds": dt,
            "ray_steps_traced": int(rep.ray_steps), "ray_steps_per_s": rep.ray_steps / dt,
            "traces_per_pass": 1, "ray_steps_final_pass": int(rep.ray_steps_final),
            "absorbed_sum": absorbed, "max_beam_gain_over_mean_absorbed": float(np.abs(bg).max() / (absorbed / len(bg))),
            "relax": gp.relax, "tolerance": gp.tolerance}


def measured_traffic(workload, variant, shard_count=1):
    """Per-launch counter values of the trace kernel from the committed PMC passes (profiles/r*/traffic.json:
    SQ_INSTS_VALU, TCC_EA0_ATOMIC, (2 x FETCH_SIZE + WRITE_SIZE) * 1 KiB, each collected in its own --pmc pass); the
    newest round's entry for this workload, the shipped kernel and this share of the work (shard_count = the number of
    ranks the bundle list is cut for; entries for K > 1 were profiled on one GPU with --shard-of K) wins.  None when
    there is none."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    for rnd in sorted(os.listdir(prof)) if os.path.isdir(prof) else []:
        path = os.path.join(prof, rnd, "traffic.json")
        if os.path.exists(path):
            for e in json.load(open(path)).get("entries", []):
                if (e.get("workload") == workload and e.get("kernel_variant") == variant and
                        e.get("shard_count", 1) == shard_count and
                        e.get("kernel") == "k_trace_window" and "SQ_INSTS_VALU_per_launch" in e):
                    best = e
    return best


VALU_CYCLES = {"ADD_F64": 5.5, "MUL_F64": 5.5, "FMA_F64": 5.5, "CVT": 5.5, "TRANS_F64": 17.0, "INT32": 2.8, "INT64": 4.9}
VALU_CYCLES_OTHER = 4.9


def valu_busy(prof, kernel_s):
    """Sum over instruction classes of count x measured SIMD cycles, over the SIMD cycles of the launch; None without the
    per-class counter pass."""
    keys = ["SQ_INSTS_VALU_%s_per_launch" % k for k in VALU_CYCLES]
    if prof is None or not all(k in prof for k in keys):
        return None
    classed = sum(prof["SQ_INSTS_VALU_%s_per_launch" % k] for k in VALU_CYCLES)
    cycles = sum(prof["SQ_INSTS_VALU_%s_per_launch" % k] * c for k, c in VALU_CYCLES.items())
    cycles += max(0.0, prof["SQ_INSTS_VALU_per_launch"] - classed) * VALU_CYCLES_OTHER
    clock = prof.get("gpu_clock_hz_under_load", CLOCK_HZ)
    return cycles / (SIMDS * clock * kernel_s)


def roofline(prof, steps_per_launch, kernel_s, tot, steps_total):
    """The roofline object of the dominant kernel (k_trace_window), kernel time measured live with HIP events.

    The path is gather / ODE / scatter with the gathers served from cache and the scatters combined in LDS, so
    the algorithmic 128 B per ray-step is NOT a lower bound on HBM traffic and an "hbm" roofline
    built on it exceeds 1 (round 1).  What binds is vector-instruction issue (the step is ~75 fp64 and ~100
    32-bit VALU instructions per wavefront), with the memory-side atomic path second; both are reported against
    their ceilings, and the measured HBM traffic against the 8 TB/s peak, from the instruction and request
    counts of the committed rocprofv3 --pmc passes (profiles/r*/traffic.json) -- counts, not times: they are
    fixed by the workload, the time is this run's.  frac = achieved / peak."""
    alg = steps_per_launch * BYTES_PER_RAY_STEP / kernel_s
    dsteps = max(1.0, tot[10].item())        # ray-steps of the diagnostic launch(es) the window counts belong to
    out = {"kernel": "k_trace_window", "kernel_ms": 1e3 * kernel_s,
           "bytes_per_ray_step": BYTES_PER_RAY_STEP, "algorithmic_GBps": alg / 1e9,
           "global_atomics_per_ray_step": tot[1].item() / dsteps,
           "lane_utilisation": dsteps / max(1.0, 64.0 * tot[5].item()),
           "window_miss_ray_step_frac": tot[4].item() / dsteps,
           "window_miss_wave_step_frac": tot[6].item() / max(1.0, tot[5].item()),
           "box_b_live_wave_step_frac": tot[7].item() / max(1.0, tot[5].item()),
           "window_moves_per_wave_step": tot[8].item() / max(1.0, tot[5].item()),
           "window_counts_from": "one un-timed launch with cbet_params.window_stats = 1 (the timed launches count ray-steps and rays only)"}
    if prof is None:   # no committed counter profile for this workload / kernel: nothing to price against
        out.update({"bound": "valu_issue", "achieved": None, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instructions/s",
                    "frac": None, "traffic": None,
                    "note": "no rocprofv3 counter profile committed for this workload; algorithmic_GBps is context only"})
        return out
    valu = prof["SQ_INSTS_VALU_per_launch"] / kernel_s
    atom = prof["TCC_EA0_ATOMIC_requests"] * 64.0 / kernel_s
    hbm = prof["hbm_bytes_per_launch"] / kernel_s
    out.update({
        "bound": "valu_issue", "achieved": valu / 1e9, "peak": VALU_ISSUE_PEAK / 1e9, "unit": "G wave-instructions/s",
        "frac": valu / VALU_ISSUE_PEAK,
        "traffic": prof["hbm_bytes_per_launch"], "traffic_source": prof["source"],
        "hbm_measured_GBps": hbm / 1e9, "hbm_peak_GBps": HBM_PEAK / 1e9, "hbm_measured_frac": hbm / HBM_PEAK,
        "secondary": {"bound": "memory_side_atomics", "achieved": atom / 1e9, "peak": ATOMIC_PEAK / 1e9, "unit": "GB/s",
                      "frac": atom / ATOMIC_PEAK, "requests_per_launch": prof["TCC_EA0_ATOMIC_requests"]},
        # the other unit at the same level of use: the per-CU LDS arrays (fp64 atomic adds of the deposit windows)
        "lds": ({"bound": "lds_array_cycles", "achieved": prof["SQ_LDS_IDX_ACTIVE_per_launch"] / kernel_s / 1e9,
                 "peak": LDS_CYCLE_PEAK / 1e9, "unit": "G CU-cycles/s",
                 "frac": prof["SQ_LDS_IDX_ACTIVE_per_launch"] / kernel_s / LDS_CYCLE_PEAK}
                if "SQ_LDS_IDX_ACTIVE_per_launch" in prof else None),
        # the HONEST vector figure: every instruction class weighted with the SIMD cycles one wave-instruction of it takes
        # (scripts/ubench/valu_rate.hip on this chip: fp64 add / mul / fma / compare / convert 5.5, fp64 rcp / sqrt 17, 32-bit
        # integer 2.8, everything else -- selects on scalar masks, moves, 64-bit integer -- 4.9), over the SIMD cycles of the
        # run at the clock the chip holds under this kernel (GRBM_GUI_ACTIVE / rocprofv3's kernel time of the same command).
        # `frac` above prices every instruction at 4 cycles and the 2.4 GHz maximum.
        "valu_busy_frac": valu_busy(prof, kernel_s),
        # how much of the issued vector work is the reference's own arithmetic: 62 fp64 instructions per wave-step (3 kick, 6
        # drift, 6 cell units, 3 + 6 relocation, 3 conversions, 12 offsets and factors, 20 products, 2 absorption, 1 energy
        # test; launch_ray_XZ.cu:268-356; the kernel itself forms the deposit with 14 products + 8 fused adds into the pending
        # sums) at 4 issue cycles each -- the rest of `frac` is index math, window logic and the flush of the sums
        "reference_arithmetic_frac": (62.0 * prof["wave_steps_per_launch"] / kernel_s / VALU_ISSUE_PEAK
                                      if "wave_steps_per_launch" in prof else None),
        # the wave's own clock (SQ_WAVE_CYCLES and its disjoint parts, quad-cycles): what a wave-step costs the wave that
        # runs it and where that time goes; the kernel retires wave-steps at (waves per CU) / cycles_per_wave_step per CU
        "wave_time": ({"cycles_per_wave_step": 4.0 * prof["SQ_WAVE_CYCLES_per_launch"] / prof["wave_steps_per_launch"],
                       "waiting_frac": prof["SQ_WAIT_ANY_per_launch"] / prof["SQ_WAVE_CYCLES_per_launch"],
                       "issue_stalled_frac": prof["SQ_WAIT_INST_ANY_per_launch"] / prof["SQ_WAVE_CYCLES_per_launch"],
                       "issuing_frac": prof["SQ_ACTIVE_INST_ANY_per_launch"] / prof["SQ_WAVE_CYCLES_per_launch"],
                       "instructions_per_wave_step": {k: prof["SQ_INSTS_%s_per_launch" % k] / prof["wave_steps_per_launch"]
                                                      for k in ("VALU", "SALU", "LDS")},
                       "note": "every vector and every scalar instruction costs its wave one quad-cycle, an LDS instruction ~4-7 "
                               "(SQ_ACTIVE_INST_* / SQ_INSTS_*); 16 waves per CU (10,240 B of LDS and 127 VGPRs a wave: both "
                               "resources' cap), vector issue ~80 % of the SIMD cycles -- DESIGN.md 4.4"}
                      if all(k in prof for k in ("SQ_WAVE_CYCLES_per_launch", "SQ_WAIT_ANY_per_launch", "SQ_WAIT_INST_ANY_per_launch",
                                                 "SQ_ACTIVE_INST_ANY_per_launch", "wave_steps_per_launch")) else None),
        "formula": "frac = SQ_INSTS_VALU / kernel_s / (1024 SIMDs x 2.4 GHz / 4); secondary.frac = TCC_EA0_ATOMIC x 64 B / "
                   "kernel_s / 1.3 TB/s; hbm_measured_frac = (2 x FETCH_SIZE + WRITE_SIZE) x 1 KiB / kernel_s / 8 TB/s (FETCH_SIZE tallies 128-B line requests at 64 B on gfx950: calibrated, profiles/r2/fetch_calibration.log)",
        "note": "bound = vector-instruction issue, with the memory-side atomic unit level with it since round 5's last kernel "
                "(valu_busy_frac against secondary.frac: whichever is larger binds, both are within a few percent); the algorithmic "
                "128 B/ray-step figure is kept as algorithmic_GBps for context -- it exceeds the HBM peak because gathers "
                "hit L2/MALL and scatters are combined in LDS (DESIGN.md 4.3)"})
    return out


def visible_gpu_count():
    """HIP devices this node offers, counted WITHOUT creating a HIP / HSA context in this process: the KFD topology in
    sysfs (nodes with simd_count > 0 are GPUs), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set.  The
    launcher process stays off the GPU entirely; the ranks check their own device ordinal again."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    if os.path.isdir(root):
        for node in os.listdir(root):
            try:
                props = dict(l.split()[:2] for l in open(os.path.join(root, node, "properties")) if len(l.split()) >= 2)
                n += 1 if int(props.get("simd_count", "0")) > 0 else 0
            except (OSError, ValueError):
                pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([x for x in os.environ[var].split(",") if x.strip()]))
    return n


def spawn_ranks(ngpus, rank_timeout):
    """Start `ngpus` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a
    free port) and return the launcher's exit code.  This process never touches the GPU (no torch import, no HIP call)
    and nothing is exec'd: the ranks are fresh children.  They run under a parent-side limit (`--rank-timeout` seconds,
    well inside the driver's): on expiry the whole process group of the launcher is terminated, every rank's last
    stderr lines are printed and the exit code is 124 -- a run that hangs in RCCL's set-up cannot die silently.  Every
    rank's stderr goes to a file of its own (CBET_BENCH_STDERR_DIR) and is replayed, labelled, when the run ends."""
    import shutil
    import signal
    import socket
    import subprocess
    import tempfile
    if "CBET_BENCH_DEVICE" not in os.environ:     # (the rehearsal override puts every rank on one device)
        have = visible_gpu_count()
        if ngpus > have:
            print("bench.py: --gpus %d but this node has %d HIP device(s)" % (ngpus, have), file=sys.stderr)
            return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("NCCL_DEBUG", "WARN")                # RCCL says why when it refuses or stalls
    logdir = tempfile.mkdtemp(prefix="cbet_bench_ranks_")
    env["CBET_BENCH_STDERR_DIR"] = logdir

    def replay(tail=None):
        for name in sorted(os.listdir(logdir)):
            lines = open(os.path.join(logdir, name), errors="replace").read().splitlines()
            for line in (lines[-tail:] if tail else lines):
                print("[%s] %s" % (name.split(".")[0], line), file=sys.stderr)
        sys.stderr.flush()

    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        rc = proc.wait(timeout=rank_timeout)
    except subprocess.TimeoutExpired:
        print("bench.py: the %d ranks did not finish within %d s (--rank-timeout): terminating them" % (ngpus, rank_timeout),
              file=sys.stderr)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)               # the launcher's own session: it and every rank, nothing else
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        replay(tail=30)
        shutil.rmtree(logdir, ignore_errors=True)
        return 124
    replay(tail=None if rc == 0 else 60)
    shutil.rmtree(logdir, ignore_errors=True)
    return rc


def dense_layout_times(api, RayTracer, SweepPipeline, p, r, ne, te, bn, samples):
    """The same trace launch into the REFERENCE's deposit-grid layout -- dense rows of nz + 2 doubles (def.cuh:121-131,
    main.cu:262, launch_ray_XZ.cu:5-7), what a caller of cbet_launch_ray_XYZ holding the reference's array gets -- timed on
    `samples` fresh (context, grid) allocations kept alive side by side: with dense rows the time depends on where the grid
    lands relative to the record table (Placement), so the spread is reported, not one number.  Outside the
    headline's timed region."""
    import torch
    keep, ms = [], []
    for _ in range(samples):
        t = RayTracer(p, r, ne, te, beam_norm=bn)
        pp = SweepPipeline(t, 0, 1, pad_rows=0)
        pp.run_pass()                       # tables and records of this context
        ms.append(1e3 * pp.time_trace_alone(reps=3))
        keep.append((t, pp))
    for t, pp in keep:
        pp.close()
        t.close()
    torch.cuda.synchronize()
    return {"edep_row_pitch": int(p.nz + 2), "samples": samples, "min_ms": min(ms), "mean_ms": sum(ms) / len(ms), "max_ms": max(ms),
            "all_ms": [round(x, 3) for x in ms],
            "note": "k_trace_window alone (HIP events, 3 launches each) into dense (n+2)^3 grids -- the reference's edep layout "
                    "-- on fresh context + grid allocations held side by side; the headline's grids have padded rows "
                    "(config.edep_row_pitch)"}


def device_identity(torch, index):
    pr = torch.cuda.get_device_properties(index)
    uuid = getattr(pr, "uuid", None)
    return {"ordinal": int(index), "uuid": str(uuid) if uuid is not None else None,
            "pci": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
            "name": pr.name}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--grid", dest="n", type=int, default=256, help="grid nodes per axis (BASELINE config 3: 256)")
    ap.add_argument("--rays-per-zone", type=int, default=4, help="def.cuh:58 ships 4; BASELINE config 5 as stated (1.13e6 ray "
                    "ids per beam at 512^3) is 6")
    ap.add_argument("--patch-order", type=int, default=None, help="cbet_params.patch_order (default: the library's)")
    ap.add_argument("--rim-merge", type=int, default=None, help="cbet_params.rim_merge in launch zones (default: the library's 4; 0 = one 8x8 patch per bundle)")
    ap.add_argument("--variant", type=int, default=0, help="cbet_params.kernel_variant (0 = default)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank flow on a 1-GPU box together with CBET_BENCH_DEVICE)")
    ap.add_argument("--shard-of", type=int, default=0, help="profiling aid, one process: trace the middle rank's share of a "
                    "K-rank run on this GPU (no process group; the combine is a local slab copy) -- how the per-share "
                    "counter profiles of profiles/r*/traffic.json are collected")
    ap.add_argument("--rank-timeout", type=int, default=420, help="--gpus N without a launcher: seconds the N ranks may take "
                    "before the parent terminates them and prints their last stderr lines (exit code 124)")
    ap.add_argument("--dense-samples", type=int, default=5, help="N = 1: fresh (context, grid) allocations the trace is timed on "
                    "with the reference's dense edep rows (reported as dense_layout, outside the timed region); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cbet", action="store_true", help="skip the (unpinned) CBET-iteration leg reported beside the headline")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: this process becomes the launcher (the counterpart of the one
        # OpenMP thread per GPU of main.cu:166-176).  It never touches the GPU -- the ranks are fresh children of
        # torch.distributed.run, no exec from a process that holds a HIP context -- and relays their output and exit code.
        sys.exit(spawn_ranks(args.gpus, args.rank_timeout))
    if os.environ.get("CBET_BENCH_STDERR_DIR") and "RANK" in os.environ:
        # a rank started by spawn_ranks: its stderr (Python's and the native libraries') goes to a file of its own
        fd = os.open(os.path.join(os.environ["CBET_BENCH_STDERR_DIR"], "rank%s.stderr" % os.environ["RANK"]),
                     os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
        os.dup2(fd, 2)
        os.close(fd)
        if os.environ.get("CBET_BENCH_TEST_HANG"):     # tests/test_bench_cli.py: a rank that never comes back
            print("rank %s: hanging on request (CBET_BENCH_TEST_HANG)" % os.environ["RANK"], file=sys.stderr, flush=True)
            time.sleep(3600)

    import torch
    import torch.distributed as dist
    from cbet_raytracing_3d_amd import api
    from cbet_raytracing_3d_amd.tracer import RayTracer, SweepPipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    device_index = int(os.environ.get("CBET_BENCH_DEVICE", local_rank))  # rehearsal override only
    if device_index >= torch.cuda.device_count():
        raise SystemExit("rank %d wants device %d but this node has %d" % (rank, device_index, torch.cuda.device_count()))
    torch.cuda.set_device(device_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:   # "n_gpus" below is what the process group really has
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    # who is in the group: every rank's device, gathered over the process group itself (distinct devices = real GPUs)
    identities = [device_identity(torch, device_index)]
    if world > 1:
        identities = [None] * world
        dist.all_gather_object(identities, device_identity(torch, device_index))

    n = args.n
    r, ne, te = api.load_s83177()
    bn = api.omega60_beam_norm()
    p = api.default_params(n, kernel_variant=args.variant, rays_per_zone=args.rays_per_zone,
                           **({} if args.patch_order is None else {"patch_order": args.patch_order}),
                           **({} if args.rim_merge is None else {"rim_merge": args.rim_merge}))
    workload = "omega60_%dcube_s83177_absorption" % n + ("" if args.rays_per_zone == 4 else "_rpz%d" % args.rays_per_zone)
    tr = RayTracer(p, r, ne, te, beam_norm=bn)
    d = tr.derived
    shards = args.shard_of if (world == 1 and args.shard_of > 1) else world      # parts the bundle list is cut into
    pipe = SweepPipeline(tr, shards // 2 if shards != world else rank, shards)

    def fence():
        pipe.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipe.warm()            # N > 1: RCCL's communicator / channel set-up, outside every pass
    for _ in range(args.warmup):
        pipe.run_pass()
    fence()
    pipe.counters(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.run_pass(timed=True)
    fence()
    elapsed = time.perf_counter() - t0

    cnt = pipe.counters(reset=True)
    slab = pipe.finish()
    slab_sum = float(slab.sum().item())
    # the deposit windows' diagnostics: ONE more launch of this rank's share, un-timed, with cbet_params.window_stats = 1
    # (the timed launches count ray-steps and rays only); tot[1], tot[4:9] and tot[10] are that launch's counts
    diag = pipe.window_diagnostics()
    tot = torch.tensor([float(cnt.ray_steps), float(diag.global_atomics), elapsed,
                        sum(a.elapsed_time(b) for a, b in pipe.kernel_events) * 1e-3, float(diag.lds_evictions),
                        float(diag.wave_steps), float(diag.wave_steps_miss), float(diag.wave_steps_wide),
                        float(diag.slabs_retired), slab_sum, float(diag.ray_steps)],
                       dtype=torch.float64, device="cuda")
    tmax = tot.clone()
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)    # tot[9]: the slabs of all ranks add up to the whole combined grid
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    edep_sum = tot[9].item()
    steps_total = tot[0].item()                      # ray-steps over all ranks and all K steps
    elapsed_max = tmax[2].item()
    kernel_s_rank = tmax[3].item() / max(1, args.steps)   # slowest rank's average trace-kernel time (stretched when traces overlap)
    value = steps_total / elapsed_max
    # what the launch needs when nothing runs beside it: the duration the roofline is priced with when traces overlap
    alone = torch.tensor([pipe.time_trace_alone() if shards > 1 else kernel_s_rank], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(alone, op=dist.ReduceOp.MAX)
    kernel_s_alone = alone.item()

    if rank == 0:
        steps_per_launch = steps_total / args.steps / world   # ray-steps one launch processes (avg rank)
        traffic = measured_traffic(workload, args.variant, shards)
        out = {
            "metric": "ray-steps/sec, OMEGA 60-beam %d^3 sweep" % n,
            "value": value, "unit": "ray-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (OMEGA-60 port table + s83177 ne/Te profile, deterministic)",
            "config": {"workload": workload, "grid": n, "beams": 60, "rays_per_zone": args.rays_per_zone,
                       "edep_sum": edep_sum, "backend": args.backend if world > 1 else None,
                       "ray_steps_per_pass": steps_total / args.steps,
                       "rays_per_pass": 60 * int(d.nlive_rays), "kernel_variant": args.variant, "rim_merge": int(p.rim_merge),
                       "edep_row_pitch": int(pipe.launch_p.edep_zpitch) or int(p.nz + 2),   # doubles per row of the timed grids (nz + 2 = the reference's dense rows)
                       "sharding": "contiguous 1/%d parts of the beam-major ray-bundle list, %s" % (shards, COMBINE_NOTE),
                       **({"shard_emulation": "rank %d of %d on one GPU, no process group (--shard-of): `value` is this "
                                              "share's rate, a profiling aid, not a bench line" % (shards // 2, shards)}
                          if shards != world else {})},
            "roofline": roofline(traffic, steps_per_launch, kernel_s_alone, tot, steps_total),
            "pipeline": {"passes_in_flight": 2, "traces_overlap": bool(pipe.overlap_traces),
                         "kernel_ms_in_pipeline": 1e3 * kernel_s_rank, "kernel_ms_alone": 1e3 * kernel_s_alone,
                         "note": "N > 1: consecutive passes' trace kernels run on separate streams and overlap (the drain of "
                                 "one beside the head of the next), so the events around a launch in the pipeline measure a "
                                 "stretched duration (kernel_ms_in_pipeline); roofline.kernel_ms is kernel_ms_alone, the "
                                 "same launch timed after the run with nothing beside it, and the counter profile is the "
                                 "one collected for this share of the work (traffic.json shard_count); N = 1 keeps one "
                                 "trace stream and the two are the same measurement"},
        }
        distinct = sorted({(d["uuid"] or d["pci"], d["ordinal"]) for d in identities})
        out["ranks"] = {"process_group_ranks": world, "backend": (args.backend if world > 1 else None),
                        "rccl_ranks": (dist.get_world_size() if (world > 1 and args.backend == "nccl") else (1 if world == 1 else 0)),
                        "distinct_devices": len(distinct), "devices": identities,
                        "rehearsal": bool("CBET_BENCH_DEVICE" in os.environ or (world > 1 and args.backend != "nccl")),
                        "note": "devices = every rank's HIP device as that rank reports it, gathered over the process group; a "
                                "rehearsal (ranks sharing one device, or gloo) is flagged and is not a multi-GPU measurement"}
        if world == 1 and shards == 1 and args.dense_samples > 0:
            out["dense_layout"] = dense_layout_times(api, RayTracer, SweepPipeline, p, r, ne, te, bn, args.dense_samples)
        if world == 1 and not args.no_cbet:
            pipe.close()
            out["cbet"] = cbet_leg(api, tr, tr.new_grid(), n)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, r, ne, te, bn)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
