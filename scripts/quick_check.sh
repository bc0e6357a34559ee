#!/bin/bash
# One GPU-box visit per kernel iteration: a parity subset, the bench line, three SQ counter passes.
# usage: quick_check.sh <tag> [pmc]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p "$OUT"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "parity_64 or truth_100 or window_kernel or nonuniform or sharding or wide_index_path or beam_resolved" > "$OUT/tests.log" 2>&1
echo "tests rc=$?"; tail -2 "$OUT/tests.log"
timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cbet --no-cpu-baseline --dense-samples 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("ms/step %.2f kernel %.2f atomics/step %.3f miss %.4f moved/wstep %.3f Bsteps %.3f" % (d["ms_per_step"], r["kernel_ms"], r["global_atomics_per_ray_step"],
      r["window_miss_ray_step_frac"], r["window_moves_per_wave_step"], r["box_b_live_wave_step_frac"]))
PY
if [ "${2:-}" = "pmc" ]; then bash scripts/pmc_quick.sh "$TAG/pmc" 2>&1 | tail -6; fi
