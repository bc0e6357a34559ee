#!/bin/bash
# Round profile, part A (a gpurun call is limited to 20 minutes: the round profile of gpu_profile_round.sh in three parts):
# bench lines, rocprofv3 kernel trace + stats of the bench command, the --pmc passes of the whole launch.
# usage: gpu_profile_round_a.sh <outdir-under-gpurun_out>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-round}; OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --grid 100 --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench_n100.json" 2>/dev/null
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --grid 512 --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench_n512.json" 2>/dev/null
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --grid 512 --rays-per-zone 6 --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench_n512_rpz6.json" 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-cbet --dense-samples 0 > "$OUT/trace.log" 2>&1
rc=$?; echo "trace rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
bash scripts/pmc.sh "$TAG/pmc" > "$OUT/pmc.log" 2>&1; echo "pmc rc=$?"
tail -3 "$OUT/pmc.log"; cat "$OUT/kernel_stats.csv" | head -5
