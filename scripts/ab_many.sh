#!/bin/bash
# usage: ab_many.sh rounds name... : the 256^3 bench (padded grids: placement-independent) with the shipped library and every
# build_alt/libcbet_<name>.so in turn, `rounds` times over
cd "${GRAFT_REPO_ROOT:-/root/repo}"
r=$1; shift
for i in $(seq 1 $r); do
  for n in base "$@"; do
    if [ "$n" = base ]; then unset CBET_LIB_PATH; else export CBET_LIB_PATH=$PWD/build_alt/libcbet_$n.so; fi
    timeout -k 10 120 python3 bench.py --steps 20 --warmup 4 --no-cbet --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-10s kernel %.3f ms step %.3f ms edep_sum %.10e' % ('$n', r['kernel_ms'], d['ms_per_step'], d['config']['edep_sum']))"
  done
done
