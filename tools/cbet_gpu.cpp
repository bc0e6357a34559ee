// cbet-gpu -- run-time-configured driver with the reference's command line and stdout contract
// (/root/reference/main.cu:234-357):
//     cbet-gpu [omp_threads] [--n N] [--gpus G] [--beams B] [--print] [--data DIR]
// Without --print it prints the four phase timers in main.cu:225-230's format; with --print it
// writes the -D PRINT text rendering of edep to stdout (what `make test` compares with truth_100).
// Links only against the C ABI (libcbet_mi355x.so).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cbet_mi355x.h"

int main(int argc, char **argv)
{
    int n = 100, gpus = 1, beams = 60;
    bool print = false;
    std::string data = "cbet_raytracing_3d_amd/data";
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--n" && i + 1 < argc) n = std::atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--beams" && i + 1 < argc) beams = std::atoi(argv[++i]);
        else if (a == "--data" && i + 1 < argc) data = argv[++i];
        else if (a == "--print") print = true;
        else if (i == 1 && std::atoi(argv[1]) > 0) { /* argv[1] = OpenMP threads in the reference (main.cu:236-242); no host loops here */ }
        else { std::fprintf(stderr, "usage: %s [omp_threads] [--n N] [--gpus G] [--beams B] [--print] [--data DIR]\n", argv[0]); return 2; }
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
    double t[4];
    cbet_counters cnt;
    if (cbet_ray_tracing(te.data(), r.data(), ne.data(), edep.data(), &p, nullptr, nullptr, gpus, t, &cnt) != CBET_OK) {
        std::fprintf(stderr, "%s\n", cbet_last_error());
        return 1;
    }
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
