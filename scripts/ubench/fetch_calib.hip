// Calibration of rocprofv3's FETCH_SIZE for the trace kernel's access pattern on gfx950 (MI355X_MICROARCH.md, HBM
// section: "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// A table of 32-byte records (1 GiB, four times the Infinity Cache) is read exactly once, each lane gathering whole
// records with two global_load_dwordx4 like k_trace_window's record gather:
//   mode 0  lane i of the grid reads record i                  (coalesced, 32 B per lane)
//   mode 1  ... record (i * 7919) mod M                        (scattered: every lane its own 128-B line)
//   mode 2  clusters of 4 lanes share a 128-B line, clusters scattered  (a bundle's rays sharing cells / z-runs)
// True bytes = M * 32 in every mode.  Run each mode under `rocprofv3 --pmc FETCH_SIZE` and compare.
// usage: fetch_calib.exe <mode>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) k_gather(const double4 *table, unsigned long long m, int mode, double *sink)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    unsigned long long r = i;
    if (mode == 1) r = (i * 7919ull) % m;
    if (mode == 2) r = (((i >> 2) * 7919ull) % (m >> 2)) * 4 + (i & 3);
    const double4 v = table[r];
    const double s = (v.x + v.y) + (v.z + v.w);
    if (s == 123.456) sink[0] = s;   // keeps the loads alive, never true
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const unsigned long long m = 1ull << 25;   // 33,554,432 records x 32 B = 1 GiB
    double4 *table;
    double *sink;
    if (hipMalloc((void **)&table, m * sizeof(double4)) != hipSuccess || hipMalloc((void **)&sink, 8) != hipSuccess) return 1;
    hipMemset(table, 0, m * sizeof(double4));
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, table, m, mode, sink);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("mode %d: %llu records, %.3f GiB true, %.3f ms, %.2f TB/s\n", mode, m, m * 32.0 / (1 << 30), ms, m * 32.0 / ms * 1e-9);
    return 0;
}
