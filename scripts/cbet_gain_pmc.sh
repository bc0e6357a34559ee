#!/bin/bash
# PMC passes for the CBET gain kernel (k_gain_field_sym) and the energy-field trace pass at 256^3 / 60 beams: separate
# rocprofv3 --pmc runs of scripts/cbet_scale.py (counters only, no tracing flags beside them).
# usage: cbet_gain_pmc.sh <outdir-under-gpurun_out>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 scripts/cbet_scale.py 256 60 > "$OUT/p$i.log" 2>&1
  rc=$?; echo "pass $i [$set] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        name = "k_gain_field_sym" if "k_gain_field_sym" in k else ("k_trace_window<16,0,2> energy-field pass" if "k_trace_windowILi16ELb0ELi2" in k or "k_trace_window<16, false, 2" in k else
                                                                  ("k_trace_window<8,0,4> four-component field pass" if "k_trace_window<8, false, 4" in k else None))
        if name:
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = "%s %s n=%d mean=%.6g min=%.6g max=%.6g" % (k, c, len(v), sum(v) / len(v), min(v), max(v))
            print(line); fo.write(line + "\n")
PY
