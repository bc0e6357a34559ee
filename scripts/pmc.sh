#!/bin/bash
# PMC passes (counters only: no tracing flags besides what --pmc implies) for one bench configuration.
# usage: pmc.sh <outdir> <bench args...>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
[ -f "$OUT/../counters_list.txt" ] || rocprofv3 -L > "$OUT/../counters_list.txt" 2>&1
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cbet --dense-samples 0 "$@" > "$OUT/p$i.log" 2>&1
  rc=$?; echo "pass $i [$set] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64
FETCH_SIZE
WRITE_SIZE
TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_trace" not in k and "k_tabulate" not in k and "k_step_table" not in k: continue
        if "k_trace" in k and ", true>(" in k: continue      # bench.py's un-timed diagnostic launch (cbet_params.window_stats): not the timed kernel
        agg["k_trace" if "k_trace" in k else ("k_step_table" if "k_step_table" in k else "k_tabulate")][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            line = "%s %s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v))
            print(line); fo.write(line + "\n")
PY
