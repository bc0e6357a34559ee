#!/bin/bash
# usage: ab.sh name [pairs=5] : interleaved bench runs of the shipped library and build_alt/libcbet_<name>.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 ${2:-5}); do
  for n in base $1; do
    if [ "$n" = base ]; then unset CBET_LIB_PATH; else export CBET_LIB_PATH=$PWD/build_alt/libcbet_$n.so; fi
    timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --no-cbet --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n kernel %.3f ms step %.3f ms' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
