#!/bin/bash
# Round profile, part B: one rank's share of a 2-, 4- and 8-rank run on this GPU (bench.py --shard-of K) -- the counter
# profiles an N-GPU bench line's roofline is priced with --, the share timing and the launch-size curve.
# usage: gpu_profile_round_b.sh <outdir-under-gpurun_out>   (the same directory as part A)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-round}; OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for K in 2 4 8; do
  bash scripts/pmc.sh "$TAG/pmc_k$K" --shard-of $K > "$OUT/pmc_k$K.log" 2>&1; rc=$?; echo "pmc --shard-of $K rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --shard-of $K --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench_share_k$K.json" 2>/dev/null
done
python3 scripts/shard_timing.py 256 > "$OUT/shard_timing.log" 2>/dev/null
python3 scripts/launch_size_curve.py > "$OUT/launch_size_curve.log" 2>/dev/null
cat "$OUT/shard_timing.log"
