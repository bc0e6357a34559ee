"""bench.py's N > 1 flow on the one-GPU box: two ranks share GPU 0 and gloo carries the combine (RCCL needs a device
per rank), so the numbers mean nothing -- what is checked is the contract: the pipelined passes of both ranks add
up to the whole workload, the reduce-scattered slabs add up to the known deposited total, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_bench_rehearsal():
    port = 29950 + (os.getpid() % 40)
    env = dict(os.environ, CBET_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--grid", "64",
           "--steps", "3", "--warmup", "1", "--no-cbet", "--no-cpu-baseline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "strong" and out["unit"] == "ray-steps/s"
    assert out["config"]["ray_steps_per_pass"] == 30712072                      # SURVEY.md 8(c), 64^3: both ranks' shares
    assert out["config"]["edep_sum"] == pytest.approx(6.1070952143e17, rel=1e-9)   # the slabs of the last pass, all ranks
    assert out["value"] > 0 and out["roofline"]["kernel_ms"] > 0
    # who was in the group: both ranks' devices as they report them; ranks sharing a device over gloo = a rehearsal, flagged
    rk = out["ranks"]
    assert rk["process_group_ranks"] == 2 and len(rk["devices"]) == 2 and rk["rehearsal"] is True
    assert rk["rccl_ranks"] == 0 and rk["distinct_devices"] == 1


def test_two_rank_line_carries_a_roofline():
    """The same rehearsal on the headline workload (256^3): the N = 2 line prices its roofline with the counter profile
    collected for ONE rank's share of a 2-rank run (profiles/r*/traffic.json, shard_count 2) and with the duration of
    that launch alone, not with the stretched duration it has while consecutive passes' traces overlap."""
    port = 29910 + (os.getpid() % 40)
    env = dict(os.environ, CBET_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--grid", "256",
           "--steps", "2", "--warmup", "1", "--no-cbet", "--no-cpu-baseline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    out = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][0])
    rl, pl = out["roofline"], out["pipeline"]
    assert out["n_gpus"] == 2 and out["config"]["ray_steps_per_pass"] == 2123497670
    assert rl["frac"] is not None and 0.05 < rl["frac"] < 1.0 and "pmc_k2" in rl["traffic_source"]
    assert rl["traffic"] is not None and rl["hbm_measured_frac"] > 0
    assert pl["traces_overlap"] and pl["kernel_ms_alone"] > 0 and rl["kernel_ms"] == pytest.approx(pl["kernel_ms_alone"])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (how the driver's scaling run invokes it; main.cu:166-176 starts its
    own per-GPU threads too): the script must start two ranks itself and rank 0 must report the process group's size."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CBET_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--grid", "64",
           "--steps", "2", "--warmup", "1", "--no-cbet", "--no-cpu-baseline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert run.returncode == 0, run.stdout[-1500:] + run.stderr[-1500:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ray_steps_per_pass"] == 30712072
    assert out["config"]["edep_sum"] == pytest.approx(6.1070952143e17, rel=1e-9)
