#!/bin/bash
# Timing-only experiment: k_trace time vs waves per SIMD, by padding the LDS allocation.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=cbet_raytracing_3d_amd/csrc
for extra in 0 3072 6144 11264; do
  lib=/tmp/libcbet_lds$extra.so
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared $( [ $extra -gt 0 ] && echo -DCBET_EXPERIMENT_EXTRA_LDS=$extra ) -I include -I $C -o $lib $C/*.hip $C/*.cpp -lrccl || exit 1
  CBET_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-cbet 2>/dev/null | \
    python3 -c "import sys,json; d=json.load(sys.stdin); lds=9856+$extra; print('extra LDS $extra B -> %d B/wave, %d waves/CU:' % (lds, 163840//lds), 'kernel_ms %.2f'%d['roofline']['kernel_ms'])"
done
