#!/bin/bash
# usage: alt_sweep.sh name... : bench the 256^3 pass with each build_alt/libcbet_<name>.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/alt
for n in "$@"; do
  CBET_LIB_PATH=$PWD/build_alt/libcbet_$n.so timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cbet --no-cpu-baseline > gpurun_out/alt/$n.json 2> gpurun_out/alt/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/alt/$n.err; continue; }
  python3 - "$n" <<'PY'
import json,sys
n=sys.argv[1]; d=json.load(open("gpurun_out/alt/%s.json"%n)); r=d["roofline"]
print("%-16s kernel %.3f ms  step %.3f ms  atomics/step %.4f  miss %.4f%%  Bsteps %.3f  edep_sum %.10e" % (n, r["kernel_ms"], d["ms_per_step"], r["global_atomics_per_ray_step"], 100*r["window_miss_ray_step_frac"], r["box_b_live_wave_step_frac"], d["config"]["edep_sum"]))
PY
done
