#!/bin/bash
# Round profile of the shipped default: bench line, rocprofv3 kernel trace + stats of the same command, the --pmc
# passes (each counter set in its own run, no tracing flags beside them), the per-rank share timing and the CBET
# stage's kernel times.  usage: gpu_profile_round.sh <outdir-under-gpurun_out>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-round}; OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --grid 100 --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench_n100.json" 2>/dev/null
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --grid 512 --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench_n512.json" 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --dense-samples 0 --no-cbet > "$OUT/trace.log" 2>&1
rc=$?; echo "trace rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
bash scripts/pmc.sh "$TAG/pmc" > "$OUT/pmc.log" 2>&1; echo "pmc rc=$?"
# one rank's share of a 2-, 4- and 8-rank run on this GPU (bench.py --shard-of K): the counter profiles an N-GPU bench
# line's roofline is priced with
for K in 2 4 8; do
  bash scripts/pmc.sh "$TAG/pmc_k$K" --shard-of $K > "$OUT/pmc_k$K.log" 2>&1; rc=$?; echo "pmc --shard-of $K rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --shard-of $K --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench_share_k$K.json" 2>/dev/null
done
python3 scripts/shard_timing.py 256 > "$OUT/shard_timing.log" 2>/dev/null
python3 scripts/launch_size_curve.py > "$OUT/launch_size_curve.log" 2>/dev/null
bash scripts/cbet_profile.sh "$TAG/cbet" > "$OUT/cbet_profile.log" 2>&1
bash scripts/cbet_gain_pmc.sh "$TAG/cbet_pmc" > "$OUT/cbet_pmc.log" 2>&1; echo "cbet pmc rc=$?"
timeout -k 10 400 python3 scripts/cbet_rank_share.py 8 256 64 2>/dev/null > "$OUT/cbet_rank_share.log"
tail -3 "$OUT/pmc.log"; cat "$OUT/shard_timing.log"
