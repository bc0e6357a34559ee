#!/bin/bash
# usage: ab_cbet.sh rounds name... : the default bench's CBET stage (first pass, energy-field pass, gain kernel, iteration) with the
# shipped library and every build_alt/libcbet_<name>.so in turn
cd "${GRAFT_REPO_ROOT:-/root/repo}"
r=$1; shift
for i in $(seq 1 $r); do
  for n in base "$@"; do
    if [ "$n" = base ]; then unset CBET_LIB_PATH; else export CBET_LIB_PATH=$PWD/build_alt/libcbet_$n.so; fi
    timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cbet']; it=c['iteration']
print('%-10s trace %.3f ms | energy-field pass %.3f ms, gain kernel %.3f ms, iteration %.2f ms; %s' % ('$n', d['roofline']['kernel_ms'], it['energy_field_pass']['kernel_ms'], it['gain_kernel']['kernel_ms'], it.get('ms', float('nan')), {k: v for k, v in c.items() if k not in ('iteration', 'parity') and not isinstance(v, dict)}))"
  done
done
