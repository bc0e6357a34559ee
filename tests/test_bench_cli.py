"""bench.py's command line, the parts that need no GPU: `--gpus N` without a launcher must start N ranks itself
(tests/test_gpu_bench_ranks.py runs that on the GPU box) and must refuse, with a non-zero exit and no JSON line, an N the
node cannot supply -- never fall back to one rank and print "n_gpus": 1."""
import os
import subprocess
import sys

from conftest import ROOT

LAUNCHER_ENV = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CBET_BENCH_DEVICE", "CBET_BENCH_TEST_HANG",
                "CBET_BENCH_STDERR_DIR")


def run_bench(*argv, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=300, cwd=ROOT, env=env)


def test_more_gpus_than_devices_is_refused():
    sys.path.insert(0, ROOT)
    import bench
    have = bench.visible_gpu_count()          # counted from sysfs: the launcher process never opens the GPU
    run = run_bench("--gpus", str(have + 2))
    assert run.returncode != 0
    assert "HIP device" in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]


def test_world_size_must_match_gpus():
    run = run_bench("--gpus", "1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert run.returncode != 0 and "WORLD_SIZE=2" in run.stderr
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]


def test_ranks_that_hang_are_terminated_with_their_last_words():
    """`--gpus N` without a launcher runs its ranks under a parent-side limit: ranks that never come back (here on request;
    on a real node: RCCL stuck in its set-up) are terminated as a process group, every rank's last stderr lines are
    printed with its name, the exit code is 124 and no JSON line appears."""
    import time
    t0 = time.time()
    run = run_bench("--gpus", "2", "--backend", "gloo", "--rank-timeout", "25", CBET_BENCH_DEVICE="0", CBET_BENCH_TEST_HANG="1")
    assert run.returncode == 124, (run.returncode, run.stderr[-1500:])
    assert time.time() - t0 < 120
    assert "did not finish within 25 s" in run.stderr
    for rank in (0, 1):
        assert "[rank%d] rank %d: hanging on request" % (rank, rank) in run.stderr, run.stderr[-1500:]
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")]
