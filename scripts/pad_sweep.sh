#!/bin/bash
# usage: pad_sweep.sh : the 256^3 bench (padded grids) with every build_alt/libcbet_pad_<YS>_<XS>.so, the shipped library between them
cd "${GRAFT_REPO_ROOT:-/root/repo}"
one() {
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 4 --no-cbet --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1 kernel %.3f ms step %.3f ms edep_sum %.10e' % (r['kernel_ms'], d['ms_per_step'], d['config']['edep_sum']))"
}
k=0
for f in build_alt/libcbet_pad_*.so; do
  if [ $((k % 3)) = 0 ]; then unset CBET_LIB_PATH; one base; fi
  k=$((k+1))
  export CBET_LIB_PATH=$PWD/$f; n=${f##*/libcbet_}; one ${n%.so}
done
unset CBET_LIB_PATH; one base
