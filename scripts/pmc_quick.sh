#!/bin/bash
# Three SQ passes only (instruction mix, stall split, LDS).  usage: pmc_quick.sh <outdir> <bench args...>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cbet --dense-samples 0 "$@" > "$OUT/p$i.log" 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done <<'SETS'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SETS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" in row["Kernel_Name"] and ", true>(" not in row["Kernel_Name"]:   # (not bench.py's un-timed diagnostic launch)
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(m):
        fo.write("%s %.6g\n" % (k, m[k]))
cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
def pct(x, units): return 100.0 * x / units / cyc if cyc else 0
print("kernel cycles (per XCD) %.4g" % cyc)
print("VALU busy  %5.1f%% of SIMD cycles" % pct(4 * m.get("SQ_ACTIVE_INST_VALU", 0), 1024))
print("LDS array  %5.1f%% of CU cycles (bank-conflict share %.0f%%), %.1f LDS cycles/instr" % (
    pct(m.get("SQ_LDS_IDX_ACTIVE", 0), 256), 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1)),
    m.get("SQ_LDS_IDX_ACTIVE", 0) / max(1, m.get("SQ_INSTS_LDS", 1))))
wc = m.get("SQ_WAVE_CYCLES", 1)
print("wave time: waitcnt %4.1f%%  issue-stall %4.1f%% (LDS part %4.1f%%)  active %4.1f%%" % (
    100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_LDS", 0) / wc,
    100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc))
ws = m.get("SQ_INSTS_VMEM_RD", 0) / 2.33   # wave-steps: 2.33 vector loads per wave-step measured at 256^3 (record gather + launch / retire reads)
print("per wave-step: VALU %.0f SALU %.0f LDS %.1f VMEM_WR %.2f" % (m.get("SQ_INSTS_VALU", 0) / ws, m.get("SQ_INSTS_SALU", 0) / ws,
      m.get("SQ_INSTS_LDS", 0) / ws, m.get("SQ_INSTS_VMEM_WR", 0) / ws))
print("atomic requests %.3g x64B = %.3g GB  L2 hit %.0f%%" % (m.get("TCC_EA0_ATOMIC_sum", 0), 64e-9 * m.get("TCC_EA0_ATOMIC_sum", 0),
      100 * m.get("TCC_HIT_sum", 0) / max(1, m.get("TCC_REQ_sum", 1))))
PY
