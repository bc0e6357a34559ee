"""The oracle is the checker, so it gets checked too: build it with AddressSanitizer +
UndefinedBehaviorSanitizer (CPU only) and run small, ragged traces through a C driver."""
import os
import subprocess

from conftest import DATA, ROOT

DRIVER = r'''
#include <stdio.h>
#include <stdlib.h>
#include "cbet_oracle.h"
static int load(const char *path, int n, double *r, double *v) {
    FILE *f = fopen(path, "r"); if (!f) return 1;
    for (int i = 0; i < n; ++i) if (fscanf(f, "%lf %lf", &r[i], &v[i]) != 2) return 2;
    fclose(f); return 0;
}
int main(int argc, char **argv) {
    double r[443], ne[443], te[443], bn[180];
    char p[1024];
    snprintf(p, sizeof p, "%s/s83177_te.txt", argv[1]); if (load(p, 443, r, te)) return 3;
    snprintf(p, sizeof p, "%s/s83177_ne.txt", argv[1]); if (load(p, 443, r, ne)) return 3;
    snprintf(p, sizeof p, "%s/omega60_beam_norm.txt", argv[1]);
    FILE *f = fopen(p, "r"); char line[256]; int k = 0;
    while (fgets(line, sizeof line, f)) { if (line[0] == '#') continue; if (sscanf(line, "%lf %lf %lf", &bn[k], &bn[k+1], &bn[k+2]) == 3) k += 3; }
    fclose(f); if (k != 180) return 4;
    const int dims[][4] = {{3,3,3,4},{4,5,3,2},{9,7,13,5},{24,24,24,4}};
    long long total = 0;
    for (int c = 0; c < 4; ++c) {
        cbet_oracle_config cfg; cbet_oracle_default_config(&cfg, dims[c][0]);
        cfg.ny = dims[c][1]; cfg.nz = dims[c][2]; cfg.rays_per_zone = dims[c][3]; cfg.nbeams = 5;
        cbet_oracle_derived d; cbet_oracle_derive(&cfg, &d);
        double *edep = calloc((size_t)d.edep_size, sizeof(double));
        double *ne3d = malloc(sizeof(double) * cfg.nx * cfg.ny * cfg.nz), *kap = malloc(sizeof(double) * cfg.nx * cfg.ny * cfg.nz);
        total += cbet_oracle_trace(&cfg, bn, r, ne, te, 0, 5, edep, 1, NULL);
        cbet_oracle_node_tables(&cfg, r, ne, te, ne3d, kap);
        double *path = malloc(sizeof(double) * 8 * d.nt);
        cbet_oracle_ray_path(&cfg, bn, r, ne, te, 2, d.nrays / 2, d.nt, path);
        cbet_oracle_write_text(edep, cfg.nx + 2, cfg.ny + 2, cfg.nz + 2, "/dev/null");
        free(path); free(kap); free(ne3d); free(edep);
    }
    printf("ray-steps %lld\n", total);
    return 0;
}
'''


def test_oracle_clean_under_asan_ubsan(tmp_path):
    src = tmp_path / "driver.c"
    src.write_text(DRIVER)
    exe = str(tmp_path / "driver")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu99", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "oracle"), str(src),
                           os.path.join(ROOT, "oracle", "cbet_oracle.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe, DATA], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ray-steps" in out.stdout and "ERROR" not in out.stderr and "runtime error" not in out.stderr
