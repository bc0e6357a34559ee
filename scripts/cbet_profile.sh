#!/bin/bash
# rocprofv3 kernel trace of the CBET stage at scale: per-kernel time of the field passes, normalise and gain kernels.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-cbet_prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 scripts/cbet_scale.py ${2:-256} ${3:-60} > "$OUT/run.log" 2>&1
rc=$?; echo "rc=$rc"; grep -v amdgpu.ids "$OUT/run.log" | tail -20
f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r["Name"]
    import re
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    short = m.group(1) if m else name.split("(")[0][-70:]
    print("%-72s calls %5s  avg %10.3f ms  total %10.1f ms  %5s%%" % (short, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
