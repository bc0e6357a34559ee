#!/bin/bash
# Timing-only experiments: price the two sources of global atomics in k_trace by compiling them out
# (results are wrong in those builds; they are built into /tmp on the GPU box and never shipped).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=cbet_raytracing_3d_amd/csrc
build() { hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared "${@:2}" -I include -I $C -o "$1" $C/*.hip $C/*.cpp -lrccl; }
build /tmp/libcbet_noflush.so -DCBET_EXPERIMENT_DROP_FLUSH_ATOMICS || exit 1
build /tmp/libcbet_nomiss.so -DCBET_EXPERIMENT_DROP_MISS_ATOMICS || exit 1
build /tmp/libcbet_noatomics.so -DCBET_EXPERIMENT_DROP_FLUSH_ATOMICS -DCBET_EXPERIMENT_DROP_MISS_ATOMICS || exit 1
for lib in "" /tmp/libcbet_noflush.so /tmp/libcbet_nomiss.so /tmp/libcbet_noatomics.so; do
  CBET_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cbet 2>/dev/null | \
    python3 -c "import sys,json; d=json.load(sys.stdin); print('lib=${lib:-shipped} (default config)', 'kernel_ms %.2f'%d['roofline']['kernel_ms'])"
done
