// cbet-gpu -- run-time-configured driver with the reference's command line and stdout contract
// (/root/reference/main.cu:234-357):
//     cbet-gpu [omp_threads] [--n N] [--gpus G] [--beams B] [--print] [--data DIR] [--cbet] [--npy PREFIX]
// Without --print it prints the four phase timers in main.cu:225-230's format; with --print it
// writes the -D PRINT text rendering of edep to stdout (what `make test` compares with truth_100).
// --cbet runs the CBET fixed-point iteration (cbet_cbet_solve; parity unpinned, no reference counterpart) on
// device 0 instead of the single reference pass.  --npy PREFIX writes what the reference's (dead) save2Hdf5 would
// (main.cu:37-94: Coordinate_x/y/z and Edepavg, [n][n][n]) plus the haloed edep as PREFIX_*.npy files.
// Links only against the C ABI (libcbet_mi355x.so).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cbet_mi355x.h"

// main.cu:321-351 + save2Hdf5 (:37-94), as .npy files
static int write_binary(const std::string &prefix, const cbet_params &p, const std::vector<double> &edep)
{
    const size_t nodes = (size_t)p.nx * p.ny * p.nz;
    std::vector<double> avg(nodes), x(nodes), y(nodes), z(nodes);
    if (cbet_edep_average(edep.data(), avg.data(), p.nx, p.ny, p.nz) != CBET_OK ||
        cbet_node_coordinates(&p, x.data(), y.data(), z.data()) != CBET_OK)
        return 1;
    const long inner[3] = {p.nx, p.ny, p.nz}, halo[3] = {p.nx + 2, p.ny + 2, p.nz + 2};
    const struct { const char *name; const double *data; const long *shape; } out[] = {
        {"Coordinate_x", x.data(), inner}, {"Coordinate_y", y.data(), inner}, {"Coordinate_z", z.data(), inner},
        {"Edepavg", avg.data(), inner}, {"edep", edep.data(), halo}};
    for (const auto &o : out)
        if (cbet_write_npy(o.data, 3, o.shape, (prefix + "_" + o.name + ".npy").c_str()) < 0) return 1;
    return 0;
}

int main(int argc, char **argv)
{
    int n = 100, gpus = 1, beams = 60;
    bool print = false, cbet = false;
    std::string data = "cbet_raytracing_3d_amd/data", npy;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--n" && i + 1 < argc) n = std::atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--beams" && i + 1 < argc) beams = std::atoi(argv[++i]);
        else if (a == "--data" && i + 1 < argc) data = argv[++i];
        else if (a == "--print") print = true;
        else if (a == "--cbet") cbet = true;
        else if (a == "--npy" && i + 1 < argc) npy = argv[++i];
        else if (i == 1 && std::atoi(argv[1]) > 0) { /* argv[1] = OpenMP threads in the reference (main.cu:236-242); no host loops here */ }
        else { std::fprintf(stderr, "usage: %s [omp_threads] [--n N] [--gpus G] [--beams B] [--print] [--data DIR] [--cbet] [--npy PREFIX]\n", argv[0]); return 2; }
    }
    cbet_params p;
    cbet_params_default(&p, n);
    p.nbeams = beams;
    cbet_derived d;
    if (cbet_derive(&p, &d) != CBET_OK) { std::fprintf(stderr, "%s\n", cbet_last_error()); return 1; }

    // main.cu:246-260: the Te file is read first, then the ne file (its radii are the ones kept)
    std::vector<double> r(p.nprofile), te(p.nprofile), ne(p.nprofile);
    if (cbet_read_profile((data + "/s83177_te.txt").c_str(), p.nprofile, r.data(), te.data()) != CBET_OK ||
        cbet_read_profile((data + "/s83177_ne.txt").c_str(), p.nprofile, r.data(), ne.data()) != CBET_OK) {
        std::fprintf(stderr, "%s\n", cbet_last_error());
        return 1;
    }
    std::vector<double> edep((size_t)d.edep_size, 0.0);  // main.cu:262
    if (cbet) {
        // the allocation / upload sequence of main.cu:136-151 through the same two helpers, then the iteration
        std::vector<double> phase_r(CBET_NPHASE), pow_r(CBET_NPHASE), bbeam(4 * (size_t)beams);
        const double *bn = cbet_omega60_beam_norm();
        cbet_host_power_table(phase_r.data(), pow_r.data());
        cbet_host_beam_trig(bn, beams, bbeam.data());
        struct Buf { void *d; const void *h; size_t bytes; } bufs[] = {
            {nullptr, te.data(), te.size() * 8}, {nullptr, r.data(), r.size() * 8}, {nullptr, ne.data(), ne.size() * 8},
            {nullptr, edep.data(), edep.size() * 8}, {nullptr, bbeam.data(), bbeam.size() * 8},
            {nullptr, bn, 3 * (size_t)beams * 8}, {nullptr, pow_r.data(), pow_r.size() * 8},
            {nullptr, phase_r.data(), phase_r.size() * 8}};
        for (auto &b : bufs)
            if (cbet_safeGPUAlloc(&b.d, b.bytes, 0) != CBET_OK || cbet_moveToAndFromGPU(b.d, (void *)b.h, b.bytes, 0) != CBET_OK) {
                std::fprintf(stderr, "%s\n", cbet_last_error());
                return 1;
            }
        cbet_gain_params g;
        cbet_gain_params_default(&g);
        cbet_cbet_report rep;
        if (cbet_cbet_solve((double *)bufs[0].d, (double *)bufs[1].d, (double *)bufs[2].d, (double *)bufs[3].d,
                            (double *)bufs[4].d, (double *)bufs[5].d, (double *)bufs[6].d, (double *)bufs[7].d, &p, &g,
                            nullptr, nullptr, nullptr, &rep) != CBET_OK ||
            cbet_moveToAndFromGPU(edep.data(), bufs[3].d, bufs[3].bytes, 0) != CBET_OK) {
            std::fprintf(stderr, "%s\n", cbet_last_error());
            return 1;
        }
        for (auto &b : bufs) cbet_gpuFree(b.d, 0);
        if (!npy.empty() && write_binary(npy, p, edep)) { std::fprintf(stderr, "%s\n", cbet_last_error()); return 1; }
        if (print) return cbet_write_text(edep.data(), p.nx + 2, p.ny + 2, p.nz + 2, nullptr) < 0;
        double sum = 0;
        for (double v : edep) sum += v;
        std::printf("cbet: passes %d converged %d gain-change %.3e energy-imbalance %.3e\nray-steps %llu (final pass %llu)  sum(edep) %.10e\n",
                    rep.passes, rep.converged, rep.change, rep.imbalance, rep.ray_steps, rep.ray_steps_final, sum);
        return 0;
    }
    double t[4];
    cbet_counters cnt;
    if (cbet_ray_tracing(te.data(), r.data(), ne.data(), edep.data(), &p, nullptr, nullptr, gpus, t, &cnt) != CBET_OK) {
        std::fprintf(stderr, "%s\n", cbet_last_error());
        return 1;
    }
    if (!npy.empty() && write_binary(npy, p, edep)) { std::fprintf(stderr, "%s\n", cbet_last_error()); return 1; }
    if (print) {
        if (cbet_write_text(edep.data(), p.nx + 2, p.ny + 2, p.nz + 2, nullptr) < 0) return 1;  // main.cu:353-355
    } else {
        auto sec = [](double s) { return (long)s; };
        auto usec = [](double s) { return (long)((s - (long)s) * 1e6); };
        std::printf("rt: Init %ld.%06ld\nTracing %ld.%06ld\nCombining %ld.%06ld\nTotal %ld.%06ld\n",  // main.cu:225-230
                    sec(t[0]), usec(t[0]), sec(t[1]), usec(t[1]), sec(t[2]), usec(t[2]), sec(t[3]), usec(t[3]));
        std::printf("ray-steps %llu  rays %llu  ray-steps/s (tracing phase) %.4g\n", cnt.ray_steps, cnt.rays_traced,
                    (double)cnt.ray_steps / t[1]);
    }
    return 0;
}
