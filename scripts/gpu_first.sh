#!/bin/bash
# First GPU pass: smoke -> parity tests -> short bench -> rocprofv3 kernel trace.
# Stops launching GPU work after any step that timed out (exit 124/137).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r1}
mkdir -p "$OUT"
export TMPDIR=/tmp
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }

timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
rc=$?; echo "smoke rc=$rc"; tail -3 "$OUT/smoke.log"; ok $rc || exit $rc

timeout -k 10 900 python -m pytest tests -m gpu -q -s -x > "$OUT/pytest.log" 2>&1
rc=$?; echo "pytest rc=$rc"; tail -25 "$OUT/pytest.log"; ok $rc || exit $rc

for v in 1 2; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --variant $v --no-cpu-baseline --no-cbet > "$OUT/bench_v$v.json" 2> "$OUT/bench_v$v.err"
  rc=$?; echo "bench v$v rc=$rc"; cat "$OUT/bench_v$v.json"; tail -3 "$OUT/bench_v$v.err"; ok $rc || exit $rc
done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --variant 2 --window 4 --no-cpu-baseline --no-cbet > "$OUT/bench_v2w4.json" 2> "$OUT/bench_v2w4.err"
rc=$?; echo "bench v2 w4 rc=$rc"; cat "$OUT/bench_v2w4.json"; ok $rc || exit $rc

timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cbet > "$OUT/prof.log" 2>&1
rc=$?; echo "rocprof rc=$rc"; tail -3 "$OUT/prof.log"
find "$OUT/prof" -name "*stats*" | head
