// Micro-benchmark: cost of LDS fp64 atomic adds on gfx950 as a function of active lanes and
// address pattern (feeds the deposit-stage design in DESIGN.md).  One wave per workgroup,
// `waves_per_cu` workgroups per CU; reports LDS cycles per wave-instruction = CU-cycles / instrs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Mode { ATOMIC_F64 = 0, ATOMIC_F32 = 1, RMW_F64 = 2, ATOMIC_U64 = 3 };

template <int MODE>
__global__ void __launch_bounds__(64) k(int iters, int active, int distinct, int stride, double *sink, long long *cycles)
{
    __shared__ double buf[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) buf[i] = 0.0;
    __syncthreads();
    // `distinct` different addresses among the active lanes, `stride` doubles apart
    const int slot = ((lane % distinct) * stride) & 2047;
    const double w = 1.0 + lane;
    long long t0 = clock64();
    if (lane < active) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = (slot + u * 64 * 0 + u) & 2047;
                if (MODE == ATOMIC_F64) __hip_atomic_fetch_add(&buf[s], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == ATOMIC_F32) __hip_atomic_fetch_add(reinterpret_cast<float *>(&buf[s]), (float)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == ATOMIC_U64) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(&buf[s]), (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else { volatile double *p = &buf[s]; *p = *p + w; }
            }
        }
    }
    __syncthreads();
    long long t1 = clock64();
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    double acc = 0;
    for (int i = lane; i < 2048; i += 64) acc += buf[i];
    sink[blockIdx.x * 64 + lane] = acc;
}

template <int MODE>
double run(int waves_per_cu, int iters, int active, int distinct, int stride, double *sink, long long *cyc, const char *name)
{
    const int blocks = 256 * waves_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, 10, active, distinct, stride, sink, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, iters, active, distinct, stride, sink, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    mean /= blocks;
    const double instrs = (double)iters * 8;
    // per-CU throughput: all waves of a CU share one LDS; wall cycles ~ ms * clock
    const double wall_cyc_per_instr_cu = (ms * 1e-3 * 2.4e9) / (instrs * waves_per_cu);
    std::printf("%-10s waves/CU %2d active %2d distinct %2d stride %3d : wave-cycles/instr %7.1f   CU-wall-cycles/instr %6.1f (@2.4GHz)  %.3f ms\n",
                name, waves_per_cu, active, distinct, stride, mean / instrs, wall_cyc_per_instr_cu, ms);
    return wall_cyc_per_instr_cu;
}

int main()
{
    double *sink;
    long long *cyc;
    hipMalloc(&sink, 256 * 16 * 64 * sizeof(double));
    hipMalloc(&cyc, 256 * 16 * sizeof(long long));
    const int iters = 20000;
    for (int wpc : {1, 8}) {
        for (int active : {64, 32, 16, 8}) {
            run<ATOMIC_F64>(wpc, iters, active, active, 1, sink, cyc, "add_f64");      // all distinct, contiguous
        }
        run<ATOMIC_F64>(wpc, iters, 64, 8, 1, sink, cyc, "add_f64");    // 8 lanes per address
        run<ATOMIC_F64>(wpc, iters, 64, 16, 1, sink, cyc, "add_f64");   // 4 lanes per address
        run<ATOMIC_F64>(wpc, iters, 64, 32, 1, sink, cyc, "add_f64");   // 2 lanes per address
        run<ATOMIC_F64>(wpc, iters, 64, 1, 1, sink, cyc, "add_f64");    // all same address
        run<ATOMIC_F64>(wpc, iters, 64, 64, 16, sink, cyc, "add_f64");  // distinct, same bank pair
        run<ATOMIC_F64>(wpc, iters, 64, 64, 9, sink, cyc, "add_f64");   // distinct, padded stride
        run<ATOMIC_U64>(wpc, iters, 64, 64, 1, sink, cyc, "add_u64");
        run<ATOMIC_U64>(wpc, iters, 64, 8, 1, sink, cyc, "add_u64");
        run<ATOMIC_F32>(wpc, iters, 64, 64, 1, sink, cyc, "add_f32");
        run<ATOMIC_F32>(wpc, iters, 64, 8, 1, sink, cyc, "add_f32");
        run<RMW_F64>(wpc, iters, 64, 64, 1, sink, cyc, "rmw_f64");
    }
    return 0;
}
