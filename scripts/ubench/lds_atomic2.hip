// Micro-benchmark 2: what sets the cost of one ds_add_f64 wave-instruction on gfx950 -- the number of active lanes,
// the number of lanes on the busiest address, or the total number of lanes that share an address with another one?
// (feeds the deposit stage of DESIGN.md 4.4: which pre-combination of lanes pays).  One wave per workgroup, 8
// workgroups per CU; prints CU-wall cycles per wave-instruction at 2.4 GHz.
//   pattern 0: `active` lanes, all distinct, contiguous
//   pattern 1: `active` lanes; the first `hot` of them share ONE address, the others are distinct
//   pattern 2: `active` lanes in groups of `hot` lanes per address (active / hot distinct addresses)
//   pattern 3: like 2, but the groups are interleaved across the wave (lane % distinct) instead of contiguous
//   pattern 4: `active` lanes spread over the wave with stride 64 / active (every k-th lane active), distinct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) k(int iters, int pattern, int active, int hot, double *sink)
{
    __shared__ double buf[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) buf[i] = 0.0;
    __syncthreads();
    bool on = lane < active;
    int slot = lane;
    if (pattern == 1) slot = lane < hot ? 0 : lane;
    if (pattern == 2) slot = lane / hot;
    if (pattern == 3) slot = lane % (active / hot);
    if (pattern == 4) { const int st = 64 / active; on = (lane % st) == 0; slot = lane / st; }
    const double w = 1.0 + lane;
    if (on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = (slot * 1 + u * 67) & 2047;
                __hip_atomic_fetch_add(&buf[s], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    double acc = 0;
    for (int i = lane; i < 2048; i += 64) acc += buf[i];
    sink[blockIdx.x * 64 + lane] = acc;
}

static double run(int pattern, int active, int hot, double *sink)
{
    const int wpc = 8, blocks = 256 * wpc, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, 10, pattern, active, hot, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, iters, pattern, active, hot, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = (ms * 1e-3 * 2.4e9) / ((double)iters * 8 * wpc);
    std::printf("pattern %d active %2d hot %2d : %6.1f CU-cycles per ds_add_f64\n", pattern, active, hot, cyc);
    return cyc;
}

int main()
{
    double *sink;
    hipMalloc(&sink, 256 * 8 * 64 * sizeof(double));
    for (int a : {64, 48, 32, 24, 16, 8, 4}) run(0, a, 1, sink);
    for (int a : {32, 16, 8}) run(4, a, 1, sink);
    for (int h : {2, 3, 4, 6, 8, 16}) run(1, 64, h, sink);
    for (int h : {2, 4, 8}) run(1, 32, h, sink);
    for (int h : {2, 4, 8, 16}) run(2, 64, h, sink);
    for (int h : {2, 4, 8}) run(2, 32, h, sink);
    for (int h : {2, 4, 8}) run(3, 64, h, sink);
    return 0;
}
