"""The kernels replace the reference's mutating-bound relocation loop
(launch_ray_XZ.cu:282-292) with a closed form (csrc/cbet_relocate.h).  Fuzz the two against each
other on the CPU, concentrating on the 0.5001 thresholds, the overlap bands and the grid faces."""
import os
import subprocess

from conftest import ROOT

SRC = r'''
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include "cbet_relocate.h"
int main() {
    std::mt19937_64 rng(83177);
    std::uniform_real_distribution<double> wide(-3.0, 3.0), tiny(-3e-4, 3e-4), ulp(-40, 40);
    const double anchors[] = {-2.5001, -2.4999, -1.5001, -1.5, -1.4999, -0.5001, -0.5, -0.4999, 0.0,
                              0.4999, 0.5, 0.5001, 1.4999, 1.5, 1.5001, 2.4999, 2.5001};
    long checked = 0, bad = 0, sure = 0;
    const int sizes[] = {3, 4, 5, 9, 12, 64, 100, 256, 1288};
    for (int n : sizes) {
        for (int c = 0; c < n; ++c) {
            if (n > 16 && c > 8 && c < n - 5 && c != n / 2) continue;
            for (int rep = 0; rep < 4000; ++rep) {
                double f;
                switch (rep & 3) {
                case 0: f = c + wide(rng); break;
                case 1: f = c + anchors[rng() % 17] + tiny(rng); break;
                case 2: f = std::nextafter(c + anchors[rng() % 17], 1e9 * (ulp(rng) > 0 ? 1 : -1)); break;
                default: { f = c + anchors[rng() % 17]; int k = (int)ulp(rng);
                           for (int i = 0; i < std::abs(k); ++i) f = std::nextafter(f, k > 0 ? 1e9 : -1e9); }
                }
                int a = cbet::relocate_loop(c, f, n), b = cbet::relocate_closed(c, f, n);
                ++checked;
                if (a != b) { if (bad++ < 10) std::printf("MISMATCH n=%d c=%d f=%.17g loop=%d closed=%d\n", n, c, f, a, b); }
                if (c >= cbet::kRelocateDeep && c <= n - 3) {   // the kernel's fast path: exact unless it says "far"
                    bool far = false;
                    int q = cbet::relocate_deep_interior(c, (double)c, f, far);
                    if (!far) { ++sure; if (q != a) { if (bad++ < 10) std::printf("FAST MISMATCH n=%d c=%d f=%.17g loop=%d fast=%d\n", n, c, f, a, q); } }
                }
            }
        }
    }
    std::printf("checked %ld sure %ld bad %ld\n", checked, sure, bad);
    return bad != 0;
}
'''


def test_closed_form_equals_reference_loop(tmp_path):
    src = tmp_path / "fuzz.cpp"
    src.write_text(SRC)
    exe = str(tmp_path / "fuzz")
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-I",
                           os.path.join(ROOT, "cbet_raytracing_3d_amd", "csrc"), str(src), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "bad 0" in out.stdout
    sure = int(out.stdout.split("sure")[1].split()[0])
    assert sure > 50000             # the fast path is exercised (it only declines rays that moved > 1.4998 cells)
