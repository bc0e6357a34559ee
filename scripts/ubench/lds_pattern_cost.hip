// Micro-benchmark 3: CU cycles of one ds_add_f64 wave-instruction for ARBITRARY lane -> LDS slot patterns.
// Reads int32 patterns [npat][64] (slot index in doubles, -1 = lane inactive) from a file, runs each one as the only
// LDS instruction of a loop (8 workgroups of one wave per CU) and prints the CU-wall cycles per instruction at 2.4 GHz.
// scripts/deposit_schemes.py --export writes pattern files from oracle ray paths (real deposit instructions of the
// 256^3 sweep under candidate lane mappings), so the scheme comparison rests on measured cycles, not on a model.
// usage: lds_pattern_cost.exe patterns.bin [names.txt]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

__global__ void __launch_bounds__(64) k(int iters, const int *__restrict__ pat, double *sink)
{
    __shared__ double buf[4096];   // 2 copies of a 2048-double image: consecutive instructions use different copies (5 waves per CU)
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) buf[i] = 0.0;
    __syncthreads();
    const int slot = pat[lane];
    const double w = 1.0 + lane;
    if (slot >= 0) {
        double *p = &buf[slot & 2047];   // (same banks in every copy: a copy is a multiple of 128 B)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) __hip_atomic_fetch_add(p + 2048 * (u & 1), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    double acc = 0;
    for (int i = lane; i < 4096; i += 64) acc += buf[i];
    sink[blockIdx.x * 64 + lane] = acc;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<int> pats;
    int v;
    while (std::fread(&v, 4, 1, f) == 1) pats.push_back(v);
    std::fclose(f);
    std::vector<std::string> names;
    if (argc > 2) {
        FILE *g = std::fopen(argv[2], "r");
        char line[256];
        while (g && std::fgets(line, sizeof line, g)) {
            std::string s(line);
            while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
            names.push_back(s);
        }
        if (g) std::fclose(g);
    }
    const int npat = (int)(pats.size() / 64);
    int *dp;
    double *sink;
    hipMalloc(&dp, pats.size() * 4);
    hipMemcpy(dp, pats.data(), pats.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&sink, 256 * 5 * 64 * sizeof(double));
    const int wpc = 5, blocks = 256 * wpc, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int p = 0; p < npat; ++p) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, 10, dp + 64 * p, sink);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, iters, dp + 64 * p, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = (ms * 1e-3 * 2.4e9) / ((double)iters * 8 * wpc);
        std::printf("%d %.2f %s\n", p, cyc, p < (int)names.size() ? names[p].c_str() : "");
    }
    return 0;
}
