#!/bin/bash
# timing-only experiment builds: bench line per alternative library in build_alt/ (results of these builds are wrong by design)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "$@"; do
  CBET_LIB_PATH=$PWD/build_alt/$lib timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cbet --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms %.2f' % d['roofline']['kernel_ms'])"
done
