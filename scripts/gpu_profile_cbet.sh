#!/bin/bash
# The CBET part of the round profile alone (scripts/gpu_profile_round.sh does the whole): kernel times of the stage, the
# --pmc passes of its two kernels, the rank-share timing, the gain kernel alone, and the default bench line (whose `cbet`
# object is priced with what this leaves in profiles/).  usage: gpu_profile_cbet.sh <outdir-under-gpurun_out>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-cbet_round}; OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
bash scripts/cbet_profile.sh "$TAG/cbet" > "$OUT/cbet_profile.log" 2>&1; echo "cbet profile rc=$?"
bash scripts/cbet_gain_pmc.sh "$TAG/cbet_pmc" > "$OUT/cbet_pmc.log" 2>&1; rc=$?; echo "cbet pmc rc=$rc"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python3 scripts/cbet_rank_share.py 8 256 64 2>/dev/null > "$OUT/cbet_rank_share.log"; echo "rank share rc=$?"
timeout -k 10 200 python3 scripts/gain_variants.py 256 60 hist > "$OUT/gain_kernel_alone.log" 2>/dev/null; echo "gain alone rc=$?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
tail -5 "$OUT/cbet_rank_share.log"; cat "$OUT/gain_kernel_alone.log"
