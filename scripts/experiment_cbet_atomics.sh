#!/bin/bash
# Timing-only: price the global atomics of the fused CBET field pass by compiling them out
# (results are wrong in those builds; built into /tmp on the GPU box and never shipped).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
C=cbet_raytracing_3d_amd/csrc
build() { hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared "${@:2}" -I include -I $C -o "$1" $C/*.hip $C/*.cpp -lrccl; }
build /tmp/libcbet_noflush.so -DCBET_EXPERIMENT_DROP_FLUSH_ATOMICS || exit 1
build /tmp/libcbet_noatomics.so -DCBET_EXPERIMENT_DROP_FLUSH_ATOMICS -DCBET_EXPERIMENT_DROP_MISS_ATOMICS || exit 1
cat > /tmp/fp.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from cbet_raytracing_3d_amd import api
from cbet_raytracing_3d_amd.tracer import RayTracer
r, ne, te = api.load_s83177()
tr = RayTracer(api.default_params(256), r, ne, te)
gp = api.default_gain_params()
tr.tabulate()
f, g = tr.new_fields(), tr.new_grid(per_beam=True)
def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); a.record(); fn(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 2
tr.counters(reset=True)
t0 = timed(lambda: tr.launch_cbet(f, gp, fields=True))
c = tr.counters(reset=True)
g.fill_(1.0)
t1 = timed(lambda: tr.launch_cbet(f, gp, fields=True, gain=g))
print("lib=%s field pass no gain %.2f ms, with gain %.2f ms; lane-atomics/ray-step %.3f, miss ray-step frac %.4f" %
      (os.environ.get("CBET_LIB_PATH") or "shipped", t0, t1, c.global_atomics / c.ray_steps, c.lds_evictions / c.ray_steps))
PY
for lib in "" /tmp/libcbet_noflush.so /tmp/libcbet_noatomics.so; do
  CBET_LIB_PATH=$lib timeout -k 10 200 python /tmp/fp.py 2>&1 | grep "^lib="
done
