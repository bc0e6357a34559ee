#!/bin/bash
# usage: ab_env.sh VAR A B [pairs=4] : interleaved default bench runs with VAR=A and VAR=B (same library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 ${4:-4}); do
  for v in "$2" "$3"; do
    env "$1=$v" timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --no-cbet --no-cpu-baseline --dense-samples 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1=$v kernel %.3f ms step %.3f ms atomics/step %.4f edep_sum %.10e' % (r['kernel_ms'], d['ms_per_step'], r['global_atomics_per_ray_step'], d['config']['edep_sum']))"
  done
done
