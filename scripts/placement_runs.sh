#!/bin/bash
# the default bench line in N fresh processes with dense and with padded rows of the pipeline's private grids, interleaved:
# every process lands its tables and grids somewhere else
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 ${1:-8}); do
  for v in 0 1; do
    CBET_PAD_ROWS=$v timeout -k 10 120 python3 bench.py --steps 20 --warmup 4 --no-cbet --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pad_rows=$v kernel %.3f ms step %.3f ms' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
