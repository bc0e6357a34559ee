#!/bin/bash
# usage: ab_trace.sh name [pairs=3] : the 256^3 bench (pipeline with padded grids: placement-independent) with the shipped library and build_alt/libcbet_<name>.so, interleaved
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 ${2:-3}); do
  for n in base $1; do
    if [ "$n" = base ]; then unset CBET_LIB_PATH; else export CBET_LIB_PATH=$PWD/build_alt/libcbet_$n.so; fi
    timeout -k 10 120 python3 bench.py --steps 20 --warmup 4 --no-cbet --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$n kernel %.3f ms step %.3f ms miss %.3f%% edep_sum %.10e' % (r['kernel_ms'], d['ms_per_step'], 100*r['window_miss_ray_step_frac'], d['config']['edep_sum']))"
  done
done
