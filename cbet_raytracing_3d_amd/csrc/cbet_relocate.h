// cbet_relocate.h -- the nearest-node update of /root/reference/launch_ray_XZ.cu:282-292 in two
// forms: the literal loop, and a branch-free closed form the tuned kernel uses.  Host+device so
// that tests/test_relocate_equivalence.py can fuzz one against the other on the CPU.
#ifndef CBET_RELOCATE_H_
#define CBET_RELOCATE_H_

#include <math.h>

#if defined(__HIPCC__)
#define CBET_HD __host__ __device__ __forceinline__
#else
#define CBET_HD static inline
#endif

namespace cbet {

// Literal: for (xx = min(n-1,c+1); xx >= max(0,c-1); --xx) c = (abs(xx-f) < 0.5001) ? xx : c;
// The loop's lower bound re-reads c as it is mutated.
CBET_HD int relocate_loop(int c, double f, int n)
{
    const double half = 0.5001;  // :132
    int q = (n - 1 < c + 1) ? n - 1 : c + 1;
    while (q >= ((0 > c - 1) ? 0 : c - 1)) {
        c = (fabs(q - f) < half) ? q : c;
        --q;
    }
    return c;
}

// Closed form of the same loop.  Candidates are visited downward from c+1; each match overwrites
// c and lowers the bound by one, so the walk is: c+1, c, then c-1 only if the running index is <= c,
// then c-2 only if c-1 matched (two neighbours match together only inside the 0.0002-wide overlap
// of their +-0.5001 bands, so the cascade cannot reach c-3).  m(q) = |q - f| < 0.5001:
//   m(c+1):  result = m(c) ? c : c+1          (c-1 cannot match when c+1 does)
//   else  :  result = m(c-1) ? (m(c-2) ? c-2 : c-1) : c      (a ray that jumped > 1 cell keeps c)
// with candidates outside [0, n-1] never visited.
CBET_HD int relocate_closed(int c, double f, int n)
{
    const double half = 0.5001;
    const double fc = (double)c;
    const bool up = (c + 1 <= n - 1) && (fabs((fc + 1.0) - f) < half);
    const bool mid = fabs(fc - f) < half;
    const bool dn1 = (c - 1 >= 0) && (fabs((fc - 1.0) - f) < half);
    const bool dn2 = (c - 2 >= 0) && (fabs((fc - 2.0) - f) < half);
    const int lower = dn1 ? (dn2 ? c - 2 : c - 1) : c;
    const int upper = mid ? c : c + 1;
    return up ? upper : lower;
}

// Exact evaluation for a DEEP INTERIOR cell, kRelocateDeep <= c <= n-3, given fc = (double)c.
// For such cells all four candidates c+1, c, c-1, c-2 exist, and every difference the reference forms,
// (double)q - f for q in {c-2 .. c+1} with |f - c| < 1.5, is exact (Sterbenz: q/2 <= f <= 2q once q >= 4),
// as is g = f - fc.  The loop above is then a function of the real number g alone:
//   up = 1-T < g < 1+T,  mid = |g| < T,  dn1 = -1-T < g < -1+T,  dn2 = -2-T < g < -2+T      (T = 0.5001)
//   up ? (mid ? c : c+1) : (dn1 ? (dn2 ? c-2 : c-1) : c)
// and for |g| < 1.4998 (so neither dn2 nor the "jumped more than a cell" case g >= 1+T can occur):
//   c+1 iff g >= T;   c-1 iff g < T-1;   c otherwise                (T - 1.0 is exact in fp64)
// -- two comparisons instead of eight, no ambiguity band.  `far` is raised when |g| >= 1.4998 (a ray
// that moved more than a cell: impossible at Courant 0.5, but then the caller must use relocate_closed).
constexpr int kRelocateDeep = 6;
CBET_HD int relocate_deep_interior(int c, double fc, double f, bool &far)
{
    const double half = 0.5001;
    const double g = f - fc;
    far = far || !(fabs(g) < 1.4998);
    return c + ((g >= half) ? 1 : 0) - ((g < (half - 1.0)) ? 1 : 0);
}

}  // namespace cbet
#endif
