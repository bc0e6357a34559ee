#!/bin/bash
# Timing-only: per-rank time of a 1/K share vs resident waves per CU (padded-LDS builds).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=cbet_raytracing_3d_amd/csrc
for extra in 0 3072 6144 11264 22528; do
  lib=/tmp/libcbet_lds$extra.so
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared $( [ $extra -gt 0 ] && echo -DCBET_EXPERIMENT_EXTRA_LDS=$extra ) -I include -I $C -o $lib $C/*.hip $C/*.cpp -lrccl || exit 1
  echo "== extra LDS $extra B -> $(( 163840 / (9856 + extra) )) waves/CU"
  CBET_LIB_PATH=$lib timeout -k 10 200 python scripts/shard_timing.py 256 1 2>/dev/null | grep shards
done
