#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=cbet_raytracing_3d_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -DCBET_EXPERIMENT_TIMELINE -I include -I $C -o /tmp/libcbet_timeline.so $C/*.hip $C/*.cpp -lrccl || exit 1
for K in "$@"; do CBET_LIB_PATH=/tmp/libcbet_timeline.so timeout -k 10 200 python scripts/experiment_timeline.py $K 2>/dev/null; done
