// cbet-ref-shaped -- the call sequence a maintainer of the reference ends up with after the change
// INTEGRATION.md section 2 describes: the reference's two helpers under their own names
// (safeGPUAlloc / moveToAndFromGPU, multi_gpu.cuh:6-7) through include/cbet_reference_api.hpp, the
// `<<<>>>` launch (main.cu:171-174) replaced by the host function launch_ray_XYZ with the same thirteen
// arguments, one upload / launch / download / host-sum per GPU as in rayTracing() (main.cu:131-210).
// It exists to prove that header against a real build and run:
//     cbet-ref-shaped [--n N] [--gpus G] [--print]
// (--print: the -D PRINT text dump, what `make test` compares with truth_100.)
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "cbet_reference_api.hpp"

namespace {

struct DeviceArrays {   // the eight device arrays of one GPU (main.cu:112-143)
    double *beam_norm = nullptr, *bbeam_norm = nullptr, *pow_r = nullptr, *phase_r = nullptr;
    double *ne = nullptr, *te = nullptr, *r = nullptr, *edep = nullptr;
};

bool ray_tracing(const std::vector<double> &te, const std::vector<double> &r, const std::vector<double> &ne,
                 std::vector<double> &edep, cbet_params params, int ngpus)
{
    params.ngpus = ngpus;   // launch b owns beams [b * nbeams/ngpus, (b+1) * nbeams/ngpus), launch_ray_XZ.cu:123
    cbet_derived d;
    if (cbet_derive(&params, &d) != CBET_OK) return false;
    const int nbeams = params.nbeams;
    std::vector<double> phase_r(CBET_NPHASE), pow_r(CBET_NPHASE), bbeam(4 * (size_t)nbeams);
    const double *beam_norm = cbet_omega60_beam_norm();
    cbet_host_power_table(phase_r.data(), pow_r.data());      // main.cu:102-110
    cbet_host_beam_trig(beam_norm, nbeams, bbeam.data());     // main.cu:121-129
    const size_t cells = (size_t)d.edep_size;

    std::vector<DeviceArrays> dev((size_t)ngpus);
    bool ok = true;
    for (int g = 0; g < ngpus && ok; ++g) {                   // main.cu:133-152
        DeviceArrays &a = dev[(size_t)g];
        ok = safeGPUAlloc((void **)&a.beam_norm, sizeof(double) * 3 * nbeams, g) &&
             safeGPUAlloc((void **)&a.bbeam_norm, sizeof(double) * 4 * nbeams, g) &&
             safeGPUAlloc((void **)&a.pow_r, sizeof(double) * CBET_NPHASE, g) &&
             safeGPUAlloc((void **)&a.phase_r, sizeof(double) * CBET_NPHASE, g) &&
             safeGPUAlloc((void **)&a.ne, sizeof(double) * ne.size(), g) &&
             safeGPUAlloc((void **)&a.te, sizeof(double) * te.size(), g) &&
             safeGPUAlloc((void **)&a.r, sizeof(double) * r.size(), g) &&
             safeGPUAlloc((void **)&a.edep, sizeof(double) * cells, g) &&
             hipMemset(a.edep, 0, sizeof(double) * cells) == hipSuccess &&   // the reference leaves it uninitialised
             moveToAndFromGPU(a.beam_norm, (void *)beam_norm, sizeof(double) * 3 * nbeams, g) &&
             moveToAndFromGPU(a.bbeam_norm, bbeam.data(), sizeof(double) * 4 * nbeams, g) &&
             moveToAndFromGPU(a.pow_r, pow_r.data(), sizeof(double) * CBET_NPHASE, g) &&
             moveToAndFromGPU(a.phase_r, phase_r.data(), sizeof(double) * CBET_NPHASE, g) &&
             moveToAndFromGPU(a.ne, (void *)ne.data(), sizeof(double) * ne.size(), g) &&
             moveToAndFromGPU(a.te, (void *)te.data(), sizeof(double) * te.size(), g) &&
             moveToAndFromGPU(a.r, (void *)r.data(), sizeof(double) * r.size(), g);
    }
    for (int g = 0; g < ngpus && ok; ++g) {                   // main.cu:166-176
        const DeviceArrays &a = dev[(size_t)g];
        ok = hipSetDevice(g) == hipSuccess;
        if (!ok) break;
        const int rc = launch_ray_XYZ(g, (unsigned)d.nindices, a.te, a.r, a.ne, a.edep, a.bbeam_norm, a.beam_norm, a.pow_r,
                                      a.phase_r, d.xconst, d.yconst, d.zconst, &params);
        if (rc != CBET_OK) { std::fprintf(stderr, "%s\n", cbet_last_error()); ok = false; break; }
        ok = hipDeviceSynchronize() == hipSuccess;
    }
    std::vector<double> part(cells);
    for (int g = 0; g < ngpus; ++g) {                         // main.cu:178-210: download, free, host sum
        DeviceArrays &a = dev[(size_t)g];
        if (ok && a.edep) {
            ok = moveToAndFromGPU(part.data(), a.edep, sizeof(double) * cells, g);
            if (ok)
                for (size_t k = 0; k < cells; ++k) edep[k] += part[k];
        }
        for (double *p : {a.beam_norm, a.bbeam_norm, a.pow_r, a.phase_r, a.ne, a.te, a.r, a.edep})
            if (p) cbet_gpuFree(p, g);
    }
    return ok;
}

}  // namespace

int main(int argc, char **argv)
{
    int n = 100, gpus = 1;
    bool print = false;
    std::string data = "cbet_raytracing_3d_amd/data";
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--n" && i + 1 < argc) n = std::atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--data" && i + 1 < argc) data = argv[++i];
        else if (a == "--print") print = true;
        else { std::fprintf(stderr, "usage: %s [--n N] [--gpus G] [--print] [--data DIR]\n", argv[0]); return 2; }
    }
    cbet_params params;
    cbet_params_default(&params, n);
    cbet_derived d;
    if (cbet_derive(&params, &d) != CBET_OK) { std::fprintf(stderr, "%s\n", cbet_last_error()); return 1; }
    std::vector<double> r(params.nprofile), te(params.nprofile), ne(params.nprofile);
    if (cbet_read_profile((data + "/s83177_te.txt").c_str(), params.nprofile, r.data(), te.data()) != CBET_OK ||
        cbet_read_profile((data + "/s83177_ne.txt").c_str(), params.nprofile, r.data(), ne.data()) != CBET_OK) {
        std::fprintf(stderr, "%s\n", cbet_last_error());
        return 1;
    }
    std::vector<double> edep((size_t)d.edep_size, 0.0);
    if (!ray_tracing(te, r, ne, edep, params, gpus)) return 1;
    if (print) return cbet_write_text(edep.data(), params.nx + 2, params.ny + 2, params.nz + 2, nullptr) < 0;
    double sum = 0;
    for (double v : edep) sum += v;
    std::printf("sum(edep) %.10e over %zu cells\n", sum, edep.size());
    return 0;
}
