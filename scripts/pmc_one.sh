#!/bin/bash
# One --pmc pass of the default bench command, k_trace kernels averaged.  usage: pmc_one.sh <outdir> "<counters>" [bench args...]
# (environment variables of the run are the caller's: export them first -- nothing but the program goes after `--`)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; CNT=$2; shift 2
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/p" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --dense-samples 0 --no-cbet "$@" > "$OUT/p.log" 2>&1
rc=$?; if [ $rc -ne 0 ]; then echo "rc=$rc"; tail -5 "$OUT/p.log"; exit $rc; fi
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" in row["Kernel_Name"] and ", true>(" not in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(out, " ".join("%s=%.6g" % (k, sum(v) / len(v)) for k, v in sorted(agg.items())))
PY
