#!/bin/bash
# the pass time against the byte offset of the step-record table inside its allocation (CBET_EXP_REC_OFFSET), grid offsets 0 and 16 KiB
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for off in 0 256 512 1024 2048 4096 8192 16384 32768 65536 131072 262144 524288 1048576; do
  echo -n "rec offset $off: "; CBET_EXP_REC_OFFSET=$off python3 scripts/placement_sweep.py 0,16,1,128 2>/dev/null | grep "round 1" | awk '{printf "%s KiB %s ms | ", $4, $9}'; echo
done
