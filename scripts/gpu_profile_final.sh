#!/bin/bash
# Final round profile of the shipped default: kernel trace + stats, then the PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-final}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-cbet > "$OUT/trace.log" 2>&1
rc=$?; echo "trace rc=$rc"; tail -2 "$OUT/trace.log"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
bash scripts/pmc.sh "${1:-final}/pmc" | tail -40
