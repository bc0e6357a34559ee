#!/bin/bash
# usage: rim_sweep.sh w... : the 256^3 pass with cbet_params.rim_merge = w launch zones (0 = one 8x8 patch per bundle)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for m in "$@"; do
  timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cbet --no-cpu-baseline --dense-samples 0 --rim-merge $m 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rim_merge $m: step %.3f ms kernel %.3f ms  atomics/step %.4f  miss %.4f%%  Bsteps %.3f  lane util %.4f  edep_sum %.10e steps %d' % (d['ms_per_step'], r['kernel_ms'], r['global_atomics_per_ray_step'], 100*r['window_miss_ray_step_frac'], r['box_b_live_wave_step_frac'], r['lane_utilisation'], d['config']['edep_sum'], d['config']['ray_steps_per_pass']))"
done
