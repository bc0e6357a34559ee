#!/bin/bash
# usage: ab_variant.sh "<bench args A>" "<bench args B>" [pairs=3] : interleaved bench runs of two argument sets (same library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 ${3:-3}); do
  for v in "$1" "$2"; do
    timeout -k 10 150 python3 bench.py --steps 20 --warmup 3 --no-cbet --no-cpu-baseline --dense-samples 0 $v 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v] kernel %.3f ms step %.3f ms atomics/step %.4f miss %.3f%% lane-util %.4f moves/wave-step %.3f edep_sum %.10e steps %d' % (r['kernel_ms'], d['ms_per_step'], r['global_atomics_per_ray_step'], 100*r['window_miss_ray_step_frac'], r['lane_utilisation'], r['window_moves_per_wave_step'], d['config']['edep_sum'], d['config']['ray_steps_per_pass']))" || echo "[$v] FAILED"
  done
done
