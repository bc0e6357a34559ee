#!/bin/bash
# usage: alt_sweep.sh name... : bench the 256^3 pass with each build_alt/libcbet_<name>.so (scripts/build_variant.py).
# Every variant first runs ONCE as its bounds-audited twin (libcbet_<name>_audit.so, -DCBET_DEBUG_BOUNDS: a bad access is
# counted and skipped instead of executed); a variant that attempted an out-of-range access, or whose audited twin is
# missing, is NOT timed.  (Round 3's `noconf` variant went straight to the plain build and ended in a GPU memory fault.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/alt
for n in "$@"; do
  if [ ! -f build_alt/libcbet_${n}_audit.so ]; then echo "$n SKIPPED: no audited twin (python scripts/build_variant.py $n)"; continue; fi
  if ! CBET_LIB_PATH=$PWD/build_alt/libcbet_${n}_audit.so timeout -k 10 300 python3 scripts/audit_variant.py > gpurun_out/alt/$n.audit 2>&1; then
    echo "$n NOT TIMED: the audited build failed or reported out-of-range accesses"; tail -4 gpurun_out/alt/$n.audit; continue; fi
  CBET_LIB_PATH=$PWD/build_alt/libcbet_$n.so timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cbet --no-cpu-baseline --dense-samples 0 > gpurun_out/alt/$n.json 2> gpurun_out/alt/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/alt/$n.err; continue; }
  python3 - "$n" <<'PY'
import json,sys
n=sys.argv[1]; d=json.load(open("gpurun_out/alt/%s.json"%n)); r=d["roofline"]
print("%-16s kernel %.3f ms  step %.3f ms  atomics/step %.4f  miss %.4f%%  Bsteps %.3f  edep_sum %.10e" % (n, r["kernel_ms"], d["ms_per_step"], r["global_atomics_per_ray_step"], 100*r["window_miss_ray_step_frac"], r["box_b_live_wave_step_frac"], d["config"]["edep_sum"]))
PY
done
