#!/bin/bash
# PMC passes for the fused CBET field pass (counters only).  usage: cbet_pmc.sh <outdir> [n] [with_gain]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 scripts/cbet_fieldpass.py "$@" > "$OUT/p$i.log" 2>&1
  rc=$?; echo "pass $i [$set] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT
GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for c, v in sorted(agg.items()):
        line = "k_trace(field pass) %s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v))
        print(line); fo.write(line + "\n")
PY
