#!/bin/bash
# the round's routine GPU check: the gpu test suite, smoke(), the default bench line.  usage: gpu_check_round.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p "$OUT"
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1; rc=$?; tail -4 "$OUT/tests.log"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; rc=$?; tail -2 "$OUT/smoke.log"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; rc=$?; tail -c 3000 "$OUT/bench.json"; exit $rc
