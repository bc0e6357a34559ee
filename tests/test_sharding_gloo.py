"""The N>1 path on CPU: world_size-2 gloo.  Each rank takes the product's shard plan
(api.shard_items = the kernel's bundle -> shard rule), traces its share with the oracle standing in
for the device, and the grids are combined with tracer.allreduce_grid -- the same call the GPU
ranks make over RCCL.  The combined grid must equal the unsharded one."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_inputs, parity_err


def _worker(rank, world, port, n, nbeams, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cbet_raytracing_3d_amd import api
        from cbet_raytracing_3d_amd.tracer import allreduce_grid, reduce_scatter_grid, shard_of_rank
        from oracle import cbet_oracle as O
        bn, r, ne, te = load_inputs()
        p = api.default_params(n, nbeams=nbeams)
        si, sc = shard_of_rank(rank, world)
        beams, ids = api.shard_items(p, nbeams, si, sc)
        cfg = O.default_config(n, nbeams=nbeams)
        e, steps = O.trace_list(cfg, bn[:nbeams].copy(), r, ne, te, beams, ids, nthreads=2)
        # (1) the slab combine bench.py uses: plane-padded grid -> this rank's x-slab of the sum
        planes = -(-e.shape[0] // world) * world
        padded = torch.zeros((planes,) + e.shape[1:], dtype=torch.float64)
        padded[: e.shape[0]] = torch.from_numpy(e)
        slab = torch.zeros((planes // world,) + e.shape[1:], dtype=torch.float64)
        assert reduce_scatter_grid(padded, slab) is None          # gloo: synchronous
        slabs = [torch.zeros_like(slab) for _ in range(world)]
        dist.all_gather(slabs, slab)
        # (2) the whole-grid combine for callers that need every cell on every rank
        t = torch.from_numpy(e.copy())
        allreduce_grid(t)
        total = torch.tensor([steps, len(ids)], dtype=torch.int64)
        dist.all_reduce(total)
        if rank == 0:
            np.save(os.path.join(out_dir, "edep.npy"), t.numpy())
            np.save(os.path.join(out_dir, "slabs.npy"), torch.cat(slabs)[: e.shape[0]].numpy())
            np.save(os.path.join(out_dir, "total.npy"), total.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_pass_equals_unsharded(tmp_path, oracle, world):
    # (8: the driver's scaling run -- 5 beams' bundles over 8 contiguous shares, 34 planes padded to 40 for the slab combine)
    n, nbeams = 32, 5
    port = 29500 + (os.getpid() % 500) + world
    mp.spawn(_worker, args=(world, port, n, nbeams, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "edep.npy")
    steps, nrays = np.load(tmp_path / "total.npy")
    bn, r, ne, te = load_inputs()
    cfg = oracle.default_config(n, nbeams=nbeams)
    want, wsteps = oracle.trace(cfg, bn[:nbeams].copy(), r, ne, te, nthreads=4)
    assert steps == wsteps
    assert parity_err(got, want) < 1e-11
    assert parity_err(np.load(tmp_path / "slabs.npy"), want) < 1e-11  # slab by slab the same sums (ring order may differ)
    from cbet_raytracing_3d_amd import api
    assert nrays == nbeams * api.derive(api.default_params(n, nbeams=nbeams)).nlive_rays


def test_allreduce_is_noop_without_process_group():
    from cbet_raytracing_3d_amd.tracer import allreduce_grid, reduce_scatter_grid, shard_of_rank
    t = torch.ones(4, dtype=torch.float64)
    assert allreduce_grid(t) is t and float(t.sum()) == 4.0
    g, sl = torch.arange(12, dtype=torch.float64).reshape(4, 3), torch.zeros(4, 3, dtype=torch.float64)
    assert reduce_scatter_grid(g, sl) is None and torch.equal(sl, g)   # one rank: the slab is the grid
    assert shard_of_rank(3, 8) == (3, 8)
    with pytest.raises(ValueError):
        shard_of_rank(8, 8)
