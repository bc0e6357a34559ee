#!/bin/bash
# Timing-only experiment: how long does k_trace take when the slab-flush atomics are compiled out?
# (results are wrong in that build; it only prices the atomics).  Builds into /tmp on the GPU box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=cbet_raytracing_3d_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -DCBET_EXPERIMENT_DROP_FLUSH_ATOMICS \
  -I include -I $C -o /tmp/libcbet_noflush.so $C/cbet_kernels.hip $C/cbet_abi.cpp $C/cbet_host.cpp $C/cbet_output.cpp -lrccl || exit 1
for lib in "" /tmp/libcbet_noflush.so; do
  CBET_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 1 --variant 3 --no-cpu-baseline 2>/dev/null | \
    python3 -c "import sys,json; d=json.load(sys.stdin); print('lib=${lib:-shipped}', 'kernel_ms %.2f'%d['roofline']['kernel_ms'], 'atomics/step %.3f'%d['roofline']['global_atomics_per_ray_step'])"
done
