"""The CBET stage off the GPU: the CPU restatement of the model, the host side of the
C ABI, and the multi-rank loop over gloo.

PARITY UNPINNED -- the reference has no CBET code (def.cuh:94-114: unused constants only).  What can be
checked is what the model promises: hooks off = the reference ray loop, exact pairwise antisymmetry of
the exchange, energy conservation in the linear limit and at the fixed point, and that sharding the
iteration over ranks changes nothing.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import NCPU, ROOT, load_inputs, parity_err

N, BEAMS = 24, [0, 16, 38, 55]


@pytest.fixture(scope="module")
def api():
    from cbet_raytracing_3d_amd import api as a
    a.lib()
    return a


@pytest.fixture(scope="module")
def case(oracle, inputs):
    bn, r, ne, te = inputs
    cfg = oracle.default_config(N, nbeams=len(BEAMS))
    ne3d, kap = oracle.node_tables(cfg, r, ne, te)
    g = oracle.gain_default()
    b = bn[BEAMS].copy()
    fields = np.stack([oracle.trace_cbet(cfg, g, b, ne3d, kap, quantity=q, per_beam=True, nthreads=NCPU)[0]
                       for q in (1, 2, 3, 4)])
    gain, change = oracle.gain_field(cfg, g, fields, ne3d, relax=1.0, nthreads=NCPU)
    return dict(cfg=cfg, g=g, bn=b, ne3d=ne3d, kap=kap, fields=fields, gain=gain, change=change)


def test_phi_is_expm1_over_x(oracle):
    for x in np.concatenate([np.linspace(-1, 1, 41), [1e-12, -1e-9, 3e-5]]):
        want = math.expm1(x) / x if x != 0 else 1.0
        assert abs(oracle.phi(x) - want) <= 4e-16 * want
    assert oracle.phi(0.0) == 1.0


def test_hooks_off_is_the_pinned_ray_loop(oracle, case):
    """quantity 0 without a gain field must be the node-table ray loop, bit for bit (serial order)."""
    c = case
    want, wsteps = oracle.trace_tables(c["cfg"], c["bn"], c["ne3d"], c["kap"], nthreads=1)
    got, steps, bg = oracle.trace_cbet(c["cfg"], c["g"], c["bn"], c["ne3d"], c["kap"], nthreads=1)
    assert steps == wsteps and np.array_equal(got, want) and not bg.any()
    zero = np.zeros((len(BEAMS),) + got.shape)
    got0, steps0, _ = oracle.trace_cbet(c["cfg"], c["g"], c["bn"], c["ne3d"], c["kap"], gain=zero, nthreads=1)
    assert steps0 == wsteps and np.array_equal(got0, want)


def test_fields_are_intensity_and_direction(oracle, case):
    c, d = case, None
    d = oracle.derive(case["cfg"])
    E = c["fields"][0]
    assert E.min() >= -1e-2 * E.max()   # the reference's deposit weights are not clamped: rare small negatives
    peak = E.max() / (2.99792458e10 * d.dt)           # energy x length / step length ~ intensity, W/cm^2
    assert 0.3e14 < peak < 4e14
    # far from the target a beam still travels along its axis: the displacement field points along -beam_norm
    for b in range(len(BEAMS)):
        D = c["fields"][1:, b].reshape(3, -1)
        tot = D.sum(axis=1)
        cosang = -np.dot(tot / np.linalg.norm(tot), c["bn"][b] / np.linalg.norm(c["bn"][b]))
        assert cosang > 0.9


def test_exchange_is_pairwise_antisymmetric(case):
    """sum_i I_i K_i = sum_ij I_i G_ij I_j = 0 in every cell (G_ij = -G_ji), to rounding."""
    E, K = case["fields"][0], case["gain"]
    assert np.abs(K).max() > 1.0
    cell = (E * K).sum(axis=0)
    assert np.abs(cell).max() <= 1e-12 * np.abs(E * K).sum(axis=0).max()
    assert case["change"][0] == case["change"][1] > 0   # from zero: |new - old| = |new|


def test_linear_limit_conserves_energy(oracle, case):
    """With the gain scaled down, what the rays gain is sum_cells K x field, beam by beam, and the beams'
    gains cancel to first order in the scale."""
    c = case
    for scale, bound in ((1e-2, 5e-2), (1e-3, 5e-3)):
        _, _, bg = oracle.trace_cbet(c["cfg"], c["g"], c["bn"], c["ne3d"], c["kap"], gain=c["gain"] * scale, nthreads=NCPU)
        assert np.abs(bg).max() > 0
        assert abs(bg.sum()) / np.abs(bg).sum() < bound
        field_level = (c["fields"][0] * c["gain"]).sum(axis=(1, 2, 3)) * scale
        assert np.abs(bg - field_level).max() < 20 * scale * np.abs(field_level).max()


def test_fixed_point_converges_and_conserves(oracle, case):
    c = case
    K, hist = None, []
    for it in range(10):
        F = np.stack([oracle.trace_cbet(c["cfg"], c["g"], c["bn"], c["ne3d"], c["kap"], gain=K, quantity=q,
                                        per_beam=True, nthreads=NCPU)[0] for q in (1, 2, 3, 4)])
        K, ch = oracle.gain_field(c["cfg"], c["g"], F, c["ne3d"], relax=1.0, gain=K, nthreads=NCPU)
        hist.append(ch[0] / ch[1])
        if hist[-1] < 1e-7:
            break
    assert hist[-1] < 1e-7 and hist[-1] < hist[1] < hist[0]
    e, steps, bg = oracle.trace_cbet(c["cfg"], c["g"], c["bn"], c["ne3d"], c["kap"], gain=K, nthreads=NCPU)
    assert np.abs(bg).max() > 1e11
    assert abs(bg.sum()) / np.abs(bg).sum() < 1e-5      # what beams gain and lose cancels at the fixed point
    e0, _ = oracle.trace_tables(c["cfg"], c["bn"], c["ne3d"], c["kap"], nthreads=NCPU)
    assert 0.9 < e.sum() / e0.sum() < 1.1 and parity_err(e, e0) > 1e-3


def test_gain_params_and_constants_host_side(api, oracle, case):
    g = api.default_gain_params()
    og = case["g"]
    for name in ("z_ion", "te_ev", "ti_ev", "mi_over_me", "iaw", "mach_r0", "mach_0", "mach_r1", "mach_1", "max_exponent"):
        assert getattr(g, name) == getattr(og, name), name
    assert (g.relax, g.tolerance, g.max_passes, g.direction_passes, g.directions_frozen) == (0.5, 1e-4, 40, 1, 0)
    p = api.default_params(N, nbeams=len(BEAMS))
    c1, cs, gc = api.gain_constants(p, g)
    assert (c1, cs, gc) == oracle.gain_constants(case["cfg"], og)
    assert 3.9e7 < cs < 4.1e7                           # def.cuh:113 "approx. 4e7 cm/s in this example"
    assert api.cbet_workspace_bytes(p) == (5 * len(BEAMS) * (N + 2) ** 3 + 2 + api.MAX_CBET_BEAMS) * 8   # four field components + gain
    for bad in (dict(max_exponent=0.0), dict(max_exponent=1.5), dict(relax=0.0), dict(relax=1.01), dict(iaw=0.0),
                dict(mach_r1=0.01), dict(direction_passes=0)):
        with pytest.raises(api.CbetError) as ei:
            api.gain_constants(p, api.default_gain_params(**bad))
        assert ei.value.code == api.EINVAL
    with pytest.raises(api.CbetError) as ei:
        api.gain_constants(api.default_params(N, nbeams=65), g)
    assert ei.value.code == api.EINVAL


def test_slab_workspace_fits_config5(api):
    """BASELINE config 5 (512^3, 60 beams, 8 ranks): the slab-owned loop's per-rank storage must fit 288 GB of HBM
    with the node tables and step records beside it; every rank holding everything would not (326 GB).  The dense
    exchange sends from and receives into the arrays themselves: no staging."""
    p = api.default_params(512)
    hsize, plane = 514 ** 3, 514 ** 2
    tables = 8 * 512 ** 3 * (2 + 4) * 2             # ne3d + kappa3d + 32-byte step records, two buffer sets
    worst = 0
    for rank in range(8):
        b = api.cbet_slab_workspace_bytes(p, 8, rank)
        nb_r = (rank + 1) * 60 // 8 - rank * 60 // 8
        planes = (rank + 1) * 514 // 8 - rank * 514 // 8
        assert b == 8 * (5 * nb_r * hsize + 5 * 60 * planes * plane + 2 + api.MAX_CBET_BEAMS)
        assert b == api.cbet_slab_workspace_bytes_parts(p, nb_r, planes, 0)
        worst = max(worst, b)
    assert worst + tables < 288e9 and worst < 88e9
    # slabs cut by gain-update work give the outer ranks more planes than 514 / 8: twice as many still fits
    assert api.cbet_slab_workspace_bytes_parts(p, 8, 130, 0) + tables < 288e9
    assert api.cbet_slab_workspace_bytes_parts(p, 8, 10, 1000) == api.cbet_slab_workspace_bytes_parts(p, 8, 10, 0) + 8000
    assert api.cbet_slab_workspace_bytes_parts(p, 61, 10, 0) == 0 and api.cbet_slab_workspace_bytes_parts(p, 8, 515, 0) == 0
    assert api.cbet_workspace_bytes(p) > 288e9       # the all-reduce loop's whole-grid arrays do not fit
    assert api.cbet_slab_workspace_bytes(p, 8, 8) == 0 and api.cbet_slab_workspace_bytes(p, 0, 0) == 0
    assert api.cbet_slab_workspace_bytes(api.default_params(256), 1, 0) == api.cbet_workspace_bytes(api.default_params(256)) + 8 * 5 * 60 * 258 ** 3


def test_balanced_slabs_rule():
    """Slabs cut by work: contiguous, covering, at least one plane per rank, near-equal summed weight."""
    from cbet_raytracing_3d_amd.tracer import _parts, balanced_slabs, gain_update_weights
    rng = np.random.default_rng(3)
    x = np.arange(258)
    w = 1.0 + 40.0 * np.exp(-((x - 129) / 40.0) ** 2) + rng.uniform(0, 0.1, 258)      # the beams cross at the centre
    for world in (1, 2, 3, 8):
        cut = balanced_slabs(w, world)
        assert cut[0][0] == 0 and cut[-1][1] == 258 and all(cut[r][1] == cut[r + 1][0] for r in range(world - 1))
        assert all(hi > lo for lo, hi in cut)
        sums = [w[lo:hi].sum() for lo, hi in cut]
        assert max(sums) - min(sums) <= 2 * w.max()
    c8 = balanced_slabs(w, 8)
    assert c8[0][1] - c8[0][0] > 2 * (c8[3][1] - c8[3][0])          # outer slabs are much wider than central ones
    assert balanced_slabs(np.zeros(10), 4) == _parts(10, 4) and balanced_slabs(np.ones(3), 5) == _parts(3, 5)
    # the cost model: a 16-cell z-run costs what its most crowded cell costs, ~ the square of the beams there
    cnt = torch.zeros((4, 3, 40), dtype=torch.int32)
    cnt[1] = 10
    cnt[2, :, 5] = 20          # one crowded cell per row: its whole run is dear, the other runs of the plane are not
    gw = gain_update_weights(cnt)
    runs = 3 * 3               # rows x z-runs of 16 (40 -> 48 padded)
    assert gw[0] == gw[3] == 8.0 * runs and gw[1] == (100 - 78.0) * runs and gw[2] == 3 * (400 - 78.0) + 3 * 2 * 8.0


class _OracleEngine:
    """cbet_fixed_point's per-rank compute with the oracle standing in for the device."""

    def __init__(self, O, api, cfg, g, bn, ne3d, kap, nbeams):
        self.O, self.api, self.cfg, self.g, self.bn, self.ne3d, self.kap, self.nb = O, api, cfg, g, bn, ne3d, kap, nbeams
        self.p = api.default_params(cfg.nx, nbeams=nbeams)
        self.gain = None
        self.edep = np.zeros(O.grid_shape(cfg))
        self.steps = 0

    def begin(self):
        self.gain = np.zeros((self.nb,) + self.O.grid_shape(self.cfg))

    def _items(self, si, sc):
        return self.api.shard_items(self.p, self.nb, si, sc)

    def field_passes(self, use_gain, si, sc, full=True):
        # full: all four fields; else the energy field alone -- the direction fields of the last full pass are kept
        # (the oracle's gain_field normalises the raw D of that pass again each time: the same k)
        qs = (1, 2, 3, 4) if full else (1,)
        F = [self.O.trace_cbet(self.cfg, self.g, self.bn, self.ne3d, self.kap, gain=self.gain if use_gain else None,
                               quantity=q, per_beam=True, nthreads=2, items=self._items(si, sc))[0] for q in qs]
        if full:
            self.fields = torch.from_numpy(np.stack(F))
        else:
            self.fields[0] = torch.from_numpy(F[0])
        return self.fields

    def update_gain(self, fields, frozen=False):
        self.gain, ch = self.O.gain_field(self.cfg, self.g, fields.numpy(), self.ne3d, relax=1.0, gain=self.gain, nthreads=2)
        return torch.tensor(ch, dtype=torch.float64)

    def deposit(self, si, sc):
        e, steps, bg = self.O.trace_cbet(self.cfg, self.g, self.bn, self.ne3d, self.kap, gain=self.gain, nthreads=2,
                                         items=self._items(si, sc))
        self.edep += e
        self.steps = steps
        return torch.from_numpy(bg)

    # the slab-owned loop (cbet_fixed_point_slabs): whole beams per rank, the gain update per x-slab, and only
    # (own beams x whole grid) + (all beams x own slab) stored -- torch tensors, filled in place by the exchange
    def begin_beams(self, b0, b1):
        gs = self.O.grid_shape(self.cfg)
        self.b0, self.b1 = b0, b1
        self.own_fields = torch.zeros((4, b1 - b0) + gs, dtype=torch.float64)
        self.gain_own = torch.zeros((b1 - b0,) + gs, dtype=torch.float64)
        self.groups_traced = 0

    def begin_slab(self, pieces):
        gs = self.O.grid_shape(self.cfg)
        self.pieces = [tuple(pc) for pc in pieces]
        self.slab_fields = [torch.zeros((4, self.nb, hi - lo) + gs[1:], dtype=torch.float64) for lo, hi in self.pieces]
        self.gain_slab = [torch.zeros((self.nb, hi - lo) + gs[1:], dtype=torch.float64) for lo, hi in self.pieces]
        self.planes = sum(hi - lo for lo, hi in self.pieces)
        self.stored = self.own_fields.numel() + self.gain_own.numel() + sum(t.numel() for t in self.slab_fields + self.gain_slab)

    def presence_counts(self):
        return (self.own_fields[0] != 0).sum(0).to(torch.int32)

    def support_mask(self):
        # the rays of the own beams traced in bookkeeping mode (absorption = 0: no ray stops before it leaves the grid)
        cfg0 = self.O.default_config(self.cfg.nx, nbeams=self.nb, absorption=0)
        gs = self.O.grid_shape(self.cfg)
        out = np.zeros((self.b1 - self.b0,) + gs, dtype=bool)
        for b in range(self.b0, self.b1):
            e, _ = self.O.trace_tables(cfg0, self.bn, self.ne3d, self.kap, beam_lo=b, beam_hi=b + 1, nthreads=2)
            out[b - self.b0] = e != 0
        return torch.from_numpy(out)

    def _beam_items(self, lo=None, hi=None):
        lo, hi = (self.b0 if lo is None else lo), (self.b1 if hi is None else hi)
        beams, ids = self.api.shard_items(self.p, self.nb, 0, 1)
        keep = (np.asarray(beams) >= lo) & (np.asarray(beams) < hi)
        return np.asarray(beams)[keep], np.asarray(ids)[keep]

    def _full_gain(self):      # the oracle takes a gain array over all beams; only this rank's beams are traced
        g = np.zeros((self.nb,) + self.O.grid_shape(self.cfg))
        g[self.b0:self.b1] = self.gain_own.numpy()
        return g

    def trace_group(self, i0, i1, use_gain, full=True, wait=()):
        # one group of this rank's beams: [b0 + i0, b0 + i1) (the device engine alternates two streams; nothing to wait for here)
        self.groups_traced += 1
        for c, q in enumerate((1, 2, 3, 4) if full else (1,)):
            F = self.O.trace_cbet(self.cfg, self.g, self.bn, self.ne3d, self.kap, gain=self._full_gain() if use_gain else None,
                                  quantity=q, per_beam=True, nthreads=2, items=self._beam_items(self.b0 + i0, self.b0 + i1))[0]
            self.own_fields[c, i0:i1] = torch.from_numpy(np.ascontiguousarray(F[self.b0 + i0:self.b0 + i1]))
        return None

    def update_gain_slab(self, frozen=False, after_piece=None):
        # the oracle updates whole grids: embed the rank's pieces (zero fields elsewhere), keep the pieces of the result.
        # after_piece(k, event) (slab_layout "halves"): piece k's new gain is stored only just before the call, so a loop that
        # sent a piece ahead of its update would send the old gain and fail the equality test
        gs = self.O.grid_shape(self.cfg)
        F = np.zeros((4, self.nb) + gs)
        old = np.zeros((self.nb,) + gs)
        for (lo, hi), f, g in zip(self.pieces, self.slab_fields, self.gain_slab):
            F[:, :, lo:hi] = f.numpy()
            old[:, lo:hi] = g.numpy()
        new, _ = self.O.gain_field(self.cfg, self.g, F, self.ne3d, relax=1.0, gain=old.copy(), nthreads=2)
        ch = [0.0, 0.0]
        for k, ((lo, hi), g) in enumerate(zip(self.pieces, self.gain_slab)):
            sl = new[:, lo:hi]
            ch[0] += np.abs(sl - g.numpy()).sum()
            ch[1] += np.abs(sl).sum()
            g.copy_(torch.from_numpy(np.ascontiguousarray(sl)))
            if after_piece is not None:
                self.pieces_announced.append(k)
                after_piece(k, None)
        return torch.tensor(ch, dtype=torch.float64)

    def deposit_beams(self):
        e, steps, bg = self.O.trace_cbet(self.cfg, self.g, self.bn, self.ne3d, self.kap, gain=self._full_gain(), nthreads=2,
                                         items=self._beam_items())
        self.edep += e
        self.steps = steps
        return torch.from_numpy(bg)


def _solve(rank, world, group=None, slabs=False, sparse=False, variant=None):
    sys.path.insert(0, ROOT)
    from cbet_raytracing_3d_amd import api
    from cbet_raytracing_3d_amd.tracer import allreduce_grid, cbet_fixed_point, cbet_fixed_point_slabs
    from oracle import cbet_oracle as O
    bn, r, ne, te = load_inputs()
    cfg = O.default_config(N, nbeams=len(BEAMS))
    ne3d, kap = O.node_tables(cfg, r, ne, te)
    eng = _OracleEngine(O, api, cfg, O.gain_default(), bn[BEAMS].copy(), ne3d, kap, len(BEAMS))
    gp = api.default_gain_params(relax=1.0, tolerance=1e-5, max_passes=8)
    if slabs:
        # (three ranks: ONE slab per rank cut by gain-update work, at most 1.5 x the equal share wide, in three trace groups;
        # else the paired layout -- two pieces per rank --, or one equal slab under the sparse plan)
        opts = dict(slab_layout=1.5, trace_groups=3) if world == 3 else dict(slab_layout="paired")
        if variant:     # "halves": the update split by plane halves; "+2ch": exchange 2 on a process group of its own
            opts = dict(slab_layout="halves", trace_groups=3 if world == 3 else 2, two_channels="2ch" in variant)
        eng.pieces_announced = []
        rep = cbet_fixed_point_slabs(eng, gp, len(BEAMS), N + 2, rank, world, group, sparse=sparse, **opts)
        eng.gain = eng.gain_own.numpy()          # this rank's beams over the whole grid
        # what the rank stored: (5 nb_r + 5 nb / W) grids = 10 nb / W: the whole problem's 5 nb at two ranks, less beyond
        full = (N + 2) ** 3
        assert eng.stored == (5 * (eng.b1 - eng.b0) * (N + 2) + 5 * len(BEAMS) * eng.planes) * (N + 2) ** 2
        assert eng.stored == 8 ** -1 * (api.cbet_slab_workspace_bytes_parts(eng.p, eng.b1 - eng.b0, eng.planes, 0) - 8 * (2 + api.MAX_CBET_BEAMS))
        cut = rep["slabs"]                     # pieces per rank: they tile the haloed grid, every plane exactly once
        assert [tuple(pc) for pc in cut[rank]] == eng.pieces
        planes = sorted(x for pcs in cut for lo, hi in pcs for x in range(lo, hi))
        assert planes == list(range(N + 2))
        assert all(len(pcs) == (2 if ((world == 2 or variant) and not sparse) else 1) for pcs in cut)
        if variant:
            assert eng.pieces_announced == [0, 1] * rep["passes"]
            assert eng.exchanger.two_channels == ("2ch" in variant and world > 1)
        assert eng.groups_traced >= rep["passes"] * min(len(rep["groups"]), 1)
        if world > 1 and not sparse:
            xch = eng.exchanger
            assert xch.plan is None and xch.chunks > 0 and xch.staging_bytes() == 0
            # dense: one message per (beam, peer, component): what this rank sent is its beams over the other ranks' slabs
            # (4 components in the direction passes, 1 afterwards) + the other ranks' beams' gain over its own slab
            npass, ndir, plane = rep["passes"], gp.direction_passes, (N + 2) ** 2
            others = (N + 2) - eng.planes
            want = 8 * plane * ((eng.b1 - eng.b0) * others * (4 * ndir + (npass - ndir)) + (len(BEAMS) - (eng.b1 - eng.b0)) * eng.planes * npass)
            assert xch.bytes_sent == want
        if world > 1 and sparse:   # only the 64-byte z-runs the rank's beams can ever touch moved
            plan = eng.exchanger.plan
            assert plan is not None and 0 < 8 * plan.runs_out < 0.6 * plan.dense_out
            npass, ndir = rep["passes"], gp.direction_passes
            assert eng.exchanger.bytes_sent == 64 * (plan.runs_out * (4 * ndir + (npass - ndir)) + plan.runs_in * npass)
    else:
        rep = cbet_fixed_point(eng, gp, rank, world, group)
    edep = torch.from_numpy(eng.edep)
    allreduce_grid(edep, group)
    steps = torch.tensor([eng.steps], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(steps)
    return rep, edep.numpy(), int(steps[0]), eng.gain


def _worker(rank, world, port, out_dir, slabs=False, sparse=False, variant=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rep, edep, steps, gain = _solve(rank, world, slabs=slabs, sparse=sparse, variant=variant)
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), edep=edep, steps=steps, gain=gain, passes=rep["passes"],
                     converged=rep["converged"], beam_gain=rep["beam_gain"], imbalance=rep["imbalance"])
    finally:
        dist.destroy_process_group()


def test_sharded_iteration_equals_unsharded(tmp_path, api, oracle):
    """world_size-2 gloo: the loop GPU ranks run over RCCL (tracer.cbet_fixed_point: shard the bundles,
    all-reduce the fields every pass, all-reduce the energy balance) against the single-rank result."""
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "out.npz")
    rep, edep, steps, gain = _solve(0, 1)
    assert rep["converged"] and bool(got["converged"]) and int(got["passes"]) == rep["passes"]
    assert int(got["steps"]) == steps
    assert parity_err(got["edep"], edep) < 1e-9
    assert np.abs(got["gain"] - gain).max() < 1e-9 * np.abs(gain).max()
    assert np.abs(got["beam_gain"] - rep["beam_gain"]).max() < 1e-9 * np.abs(rep["beam_gain"]).max()
    assert float(got["imbalance"]) < 1e-3 and rep["imbalance"] < 1e-3


def test_two_mirror_beams_exchange_nothing_net(oracle, inputs):
    """SURVEY 8(f) f1's suggested check: two beams that are mirror images of each other in a mirror-symmetric
    plasma and flow must end up with equal energy, so neither gains at the other's expense.  On a grid the
    symmetry holds to O(dx) (launch coordinates, nearest-node sampling and the deposit stencil are not mirror
    symmetric), so the net gain shrinks against the locally exchanged energy as the grid is refined; the two
    gains cancel to rounding at any resolution."""
    _, r, ne, te = inputs
    b = np.array([[0.35, 0.25, 0.90], [0.35, 0.25, -0.90]])
    b /= np.linalg.norm(b, axis=1)[:, None]
    g = oracle.gain_default()
    ratio = {}
    for n in (32, 48):
        cfg = oracle.default_config(n, nbeams=2)
        ne3d, kap = oracle.node_tables(cfg, r, ne, te)
        K = None
        for it in range(10):
            F = np.stack([oracle.trace_cbet(cfg, g, b, ne3d, kap, gain=K, quantity=q, per_beam=True, nthreads=NCPU)[0]
                          for q in (1, 2, 3, 4)])
            K, ch = oracle.gain_field(cfg, g, F, ne3d, relax=1.0, gain=K, nthreads=NCPU)
            if ch[0] / ch[1] < 1e-8:
                break
        _, _, bg = oracle.trace_cbet(cfg, g, b, ne3d, kap, gain=K, nthreads=NCPU)
        exchanged = np.abs(F[0] * K).sum(axis=(1, 2, 3))
        assert exchanged[0] > 1e12 and abs(exchanged[0] / exchanged[1] - 1) < 1e-9   # the same energy changes hands both ways
        assert abs(bg.sum()) < 1e-6 * np.abs(bg).sum()                                # conservation
        ratio[n] = np.abs(bg).max() / exchanged.max()
    assert ratio[32] < 0.15 and ratio[48] < 0.02 and ratio[48] < ratio[32]


@pytest.mark.parametrize("world,sparse,variant", [(2, False, None), (3, False, None), (2, True, None),
                                                  (2, False, "halves"), (3, False, "halves+2ch"), (2, False, "halves+2ch")])
def test_slab_owned_iteration_equals_unsharded(tmp_path, api, oracle, world, sparse, variant):
    """tracer.cbet_fixed_point_slabs over gloo: whole beams per rank, the gain update per x-slab, two point-to-point
    exchanges per pass instead of the all-reduce of every beam's fields -- same passes, same result; with the dense
    chunked exchanges (the default) and with the sparse ones (only the z-runs inside the beams' footprints move); with the
    update split by plane halves (the lower half's gain sent while the upper updates) and with exchange 2 on a second
    process group."""
    port = 29700 + (os.getpid() % 250) + world + (7 if sparse else 0) + (11 * len(variant) if variant else 0)
    mp.spawn(_worker, args=(world, port, str(tmp_path), True, sparse, variant), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    rep, edep, steps, gain = _solve(0, 1)
    assert rep["converged"] and bool(got["converged"]) and int(got["passes"]) == rep["passes"]
    assert int(got["steps"]) == steps
    assert parity_err(got["edep"], edep) < 1e-9
    b0, b1 = 0, (len(BEAMS)) // world        # rank 0 holds the gain of its own beams over the whole grid
    assert got["gain"].shape[0] == b1 - b0
    if not sparse:     # (the sparse exchange delivers a beam's gain inside its footprint only -- all its rays can read)
        assert np.abs(got["gain"] - gain[b0:b1]).max() < 1e-9 * np.abs(gain).max()
    assert np.abs(got["beam_gain"] - rep["beam_gain"]).max() < 1e-9 * np.abs(rep["beam_gain"]).max()


def test_exchanger_messages_are_contiguous_views_grouped_per_beam():
    """tracer._SlabExchanger without a process group, its transport replaced by a recorder (the hook scripts/cbet_rank_share.py
    uses): for rank 1 of 3 every grouped call must hold the messages of BOTH peers of one beam index, every message must be
    a contiguous view of the array it leaves or lands in (nothing is staged), and the own part must be copied locally."""
    from cbet_raytracing_3d_amd.tracer import _SlabExchanger, _parts
    nb, X, Y, Z, W, rank = 7, 10, 3, 4, 3, 1
    beams, slabs = _parts(nb, W), _parts(X, W)
    (b0, b1), (x0, x1) = beams[rank], slabs[rank]
    own = torch.arange(4 * (b1 - b0) * X * Y * Z, dtype=torch.float64).view(4, b1 - b0, X, Y, Z)
    slab = [torch.zeros(4, nb, x1 - x0, Y, Z, dtype=torch.float64)]
    calls = []
    xch = _SlabExchanger("cpu", None, rank, W, beams, emulate=lambda x, s, r: calls.append((s, r)))
    xch.set_slabs([[pc] for pc in slabs])
    imax = max(q1 - q0 for q0, q1 in beams)
    assert xch.fields_out(own, slab, 0, imax, range(4)) is None          # no device: no event
    assert len(calls) == imax and xch.chunks == imax
    for i, (sends, recvs) in enumerate(calls):
        mine = i < b1 - b0
        assert sorted({p for _, p in sends}) == ([0, 2] if mine else [])
        assert len(sends) == (8 if mine else 0)                           # 2 peers x 4 components
        for t, peer in sends:
            xs0, xs1 = slabs[peer]
            assert t.is_contiguous() and t.shape == (xs1 - xs0, Y, Z) and t.untyped_storage().data_ptr() == own.untyped_storage().data_ptr()
        for t, peer in recvs:
            q0, q1 = beams[peer]
            assert i < q1 - q0 and t.is_contiguous() and t.shape == (x1 - x0, Y, Z)
            assert t.untyped_storage().data_ptr() == slab[0].untyped_storage().data_ptr()
    assert torch.equal(slab[0][:, b0:b1], own[:, :, x0:x1])               # the own part never travels
    assert xch.bytes_sent == 8 * 4 * (b1 - b0) * (X - (x1 - x0)) * Y * Z and xch.staging_bytes() == 0
    # exchange 2: the gain of the peers' beams over my slab out, my beams' gain over their slabs in
    calls.clear()
    gain_slab = torch.arange(nb * (x1 - x0) * Y * Z, dtype=torch.float64).view(nb, x1 - x0, Y, Z)
    gain_own = torch.zeros(b1 - b0, X, Y, Z, dtype=torch.float64)
    xch.gain_back([gain_slab], gain_own, 0, imax)
    assert len(calls) == imax
    sent = sum(t.numel() for s_, _ in calls for t, _ in s_)
    assert sent == (nb - (b1 - b0)) * (x1 - x0) * Y * Z
    assert torch.equal(gain_own[:, x0:x1], gain_slab[b0:b1])
    # the paired layout: two pieces per rank, twice the messages, every plane owned once
    from cbet_raytracing_3d_amd.tracer import slab_pieces
    pcs = slab_pieces("paired", X, W)
    assert pcs == [[(0, 1), (5, 6)], [(1, 3), (6, 8)], [(3, 5), (8, 10)]]
    calls.clear()
    xch2 = _SlabExchanger("cpu", None, rank, W, beams, emulate=lambda x, s, r: calls.append((s, r)))
    xch2.set_slabs(pcs)
    slab2 = [torch.zeros(4, nb, hi - lo, Y, Z, dtype=torch.float64) for lo, hi in pcs[rank]]
    xch2.fields_out(own, slab2, 0, 1, range(1))
    (sends, recvs), = calls
    assert len(sends) == 4 and len(recvs) == 4                             # 2 peers x 2 pieces x 1 component, each way
    assert torch.equal(slab2[1][0, b0], own[0, 0, 6:8])
    assert slab_pieces("equal", X, W) == [[pc] for pc in slabs] and slab_pieces(1.0, X, W) == [[pc] for pc in slabs]
