// Micro-benchmark: issue rate of the VALU instructions the trace kernel's hot loop uses, on gfx950.
// 256 threads x (4 waves per SIMD) per CU; each wave runs `iters` x 8 independent copies of one
// instruction (inline asm, so the compiler cannot fold them); reports SIMD-cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) k(int iters, double *sink)
{
    double d[8];
    int i32[8];
    for (int u = 0; u < 8; ++u) { d[u] = 1.0 + threadIdx.x * 1e-3 + u; i32[u] = threadIdx.x + u; }
    const double c = 1.0000001;
    const int ci = 3;
    unsigned long long smask = 0x5555555555555555ull + iters, sm2 = 0;
    int sr = 0;
    int ci2 = 5;
    for (int it = 0; it < iters; ++it) {
#define BODY(u)                                                                                            \
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[u]) : "v"(c));                        \
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[u]) : "v"(c));                            \
        if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[u]) : "v"(c));                            \
        if (OP == 3) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[u]) : "v"(i32[u]));                       \
        if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i32[u]) : "v"(d[u]));                       \
        if (OP == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i32[u]) : "v"(ci));                      \
        if (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(i32[u]) : "v"(ci));                     \
        if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i32[u]) : "v"(ci));                \
        if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i32[u]) : "v"(ci));                         \
        if (OP == 9) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[u]), "v"(c) : "vcc");                \
        if (OP == 10) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[u]));                                        \
        if (OP == 11) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[u]));                                       \
        if (OP == 12) asm volatile("v_mov_b64 %0, %1" : "=v"(d[u]) : "v"(c));                               \
        if (OP == 13) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(d[u]) : "v"(c));                   \
        if (OP == 14) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[u]) : "v"(i32[u]), "v"(ci) : "vcc"); \
        if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(i32[u]) : "v"(ci));                            \
        if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(i32[u]) : "v"(ci), "s"(smask)); \
        if (OP == 17) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i32[u]) : "v"(ci) : "vcc");  \
        if (OP == 18) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(i32[u]), "v"(ci) : "vcc");            \
        if (OP == 19) asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(sm2) : "v"(d[u]), "v"(c));          \
        if (OP == 20) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(i32[u]) : "v"(ci), "v"(i32[(u + 1) & 7])); \
        if (OP == 21) asm volatile("v_max_i32 %0, %0, %1" : "+v"(i32[u]) : "v"(ci));                        \
        if (OP == 22) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[u]) : "v"(c));                           \
        if (OP == 23) asm volatile("v_and_b32 %0, %0, %1" : "+v"(i32[u]) : "v"(ci));                        \
        if (OP == 24) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sr) : "v"(i32[u]));                    \
        if (OP == 25) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %1, vcc" : "+v"(i32[u]), "+v"(i32[(u + 4) & 7]) : "v"(ci)); \
        if (OP == 26) asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(i32[u]) : "s"(smask));           \
        if (OP == 27) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(i32[u]) : "v"(ci));                \
        if (OP == 28) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(i32[u]) : "v"(ci));                   \
        if (OP == 29) asm volatile("v_subbrev_co_u32 %0, vcc, 0, %0, vcc" : "+v"(i32[u]) : : "vcc");       \
        if (OP == 30) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(i32[u]) : "v"(ci) : "vcc"); \
        if (OP == 31) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %1, vcc" : "+v"(i32[u]), "+v"(i32[(u + 4) & 7]) : "v"(ci) : "vcc"); \
        if (OP == 32) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(i32[u]) : "v"(d[u]), "v"(c), "v"(ci) : "vcc"); \
        if (OP == 33) asm volatile("v_cmp_lt_i32_e64 %2, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(i32[u]), "+v"(ci2), "+s"(smask) : ); \
        if (OP == 34) asm volatile("s_mov_b64 vcc, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(i32[u]) : "v"(ci), "s"(smask) : "vcc"); \
        if (OP == 35) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(i32[u]), "+v"(i32[(u + 4) & 7]) : "v"(ci) : "vcc");
        REP8(BODY)
#undef BODY
    }
    double acc = 0;
    for (int u = 0; u < 8; ++u) acc += d[u] + i32[u];
    acc += (double)sm2 + sr;
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
void run(const char *name, double *sink)
{
    const int blocks = 256 * 4, iters = 20000;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, 100, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double wave_instr_per_simd = (double)iters * 8 * 4;  // 4 waves per SIMD
    const double cycles = ms * 1e-3 * clk_khz * 1e3;
    printf("%-32s %.2f SIMD-cycles per wave-instruction (%.3f ms, clock %d MHz)\n", name, cycles / wave_instr_per_simd, ms, clk_khz / 1000);
}

int main()
{
    double *sink;
    hipMalloc(&sink, sizeof(double) * 256 * 4 * 256);
    run<0>("v_fma_f64", sink);
    run<1>("v_add_f64", sink);
    run<2>("v_mul_f64", sink);
    run<3>("v_cvt_f64_i32", sink);
    run<4>("v_cvt_i32_f64", sink);
    run<5>("v_mul_lo_u32", sink);
    run<6>("v_mul_u32_u24", sink);
    run<7>("v_cndmask_b32", sink);
    run<8>("v_add_u32", sink);
    run<9>("v_cmp_lt_f64", sink);
    run<10>("v_rcp_f64", sink);
    run<11>("v_sqrt_f64", sink);
    run<12>("v_mov_b64", sink);
    run<13>("v_lshl_add_u64", sink);
    run<14>("v_mad_u64_u32", sink);
    run<15>("v_mov_b32", sink);
    run<16>("v_cndmask_b32 e64 sgpr-mask", sink);
    run<17>("v_addc_co_u32", sink);
    run<18>("v_cmp_lt_i32 vcc", sink);
    run<19>("v_cmp_lt_f64 e64->sgpr", sink);
    run<20>("v_bfi_b32", sink);
    run<21>("v_max_i32", sink);
    run<22>("v_max_f64", sink);
    run<23>("v_and_b32", sink);
    run<24>("v_readlane_b32", sink);
    run<25>("2x v_cndmask (pair)", sink);
    run<26>("v_cndmask e64 0,1,sgpr", sink);
    run<27>("v_lshl_add_u32", sink);
    run<28>("v_add3_u32", sink);
    run<29>("v_subbrev_co_u32", sink);
    run<30>("cmp_i32 vcc + cndmask (2)", sink);
    run<31>("cmp_i32 vcc + 2 cndmask (3)", sink);
    run<32>("cmp_f64 vcc + cndmask (2)", sink);
    run<33>("cmp e64 sgpr + cndmask e64 (2)", sink);
    run<34>("s_mov vcc + cndmask (2)", sink);
    run<35>("cmp vcc, add, cndmask (3)", sink);
    return 0;
}
