#!/bin/bash
# Round profile, part C: the CBET stage -- kernel times (rocprofv3 trace of scripts/cbet_scale.py), the --pmc passes of its
# three kernels, one rank's share of the 8-rank loop.
# usage: gpu_profile_round_c.sh <outdir-under-gpurun_out>   (the same directory as part A)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-round}; OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
bash scripts/cbet_profile.sh "$TAG/cbet" > "$OUT/cbet_profile.log" 2>&1; echo "cbet profile rc=$?"
bash scripts/cbet_gain_pmc.sh "$TAG/cbet_pmc" > "$OUT/cbet_pmc.log" 2>&1; echo "cbet pmc rc=$?"
timeout -k 10 400 python3 scripts/cbet_rank_share.py 8 256 64 2>/dev/null > "$OUT/cbet_rank_share.log"
tail -5 "$OUT/cbet_profile.log"; tail -3 "$OUT/cbet_rank_share.log"
