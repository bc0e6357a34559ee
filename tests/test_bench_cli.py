"""bench.py's command line, the parts that need no GPU: `--gpus N` without a launcher must start N ranks itself
(tests/test_gpu_bench_ranks.py runs that on the GPU box) and must refuse, with a non-zero exit and no JSON line, an N the
node cannot supply -- never fall back to one rank and print "n_gpus": 1."""
import os
import subprocess
import sys

from conftest import ROOT

LAUNCHER_ENV = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CBET_BENCH_DEVICE")


def run_bench(*argv, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=300, cwd=ROOT, env=env)


def test_more_gpus_than_devices_is_refused():
    import torch
    have = torch.cuda.device_count()
    run = run_bench("--gpus", str(have + 2))
    assert run.returncode != 0
    assert "HIP device" in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]


def test_world_size_must_match_gpus():
    run = run_bench("--gpus", "1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert run.returncode != 0 and "WORLD_SIZE=2" in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]
