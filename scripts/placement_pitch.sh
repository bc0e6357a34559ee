#!/bin/bash
# the default bench line in fresh processes for a list of row pitches of the pipeline's private grids (CBET_PAD_ROWS: 0 = dense 258)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 ${1:-4}); do
  for v in 0 260 264 266 272 288 320; do
    CBET_PAD_ROWS=$v timeout -k 10 120 python3 bench.py --steps 12 --warmup 3 --no-cbet --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pitch %4s kernel %.3f ms step %.3f ms edep_sum %.10e' % ('$v', d['roofline']['kernel_ms'], d['ms_per_step'], d['config']['edep_sum']))"
  done
done
