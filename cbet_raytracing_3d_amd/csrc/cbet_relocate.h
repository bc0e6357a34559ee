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

// Fast evaluation for an INTERIOR cell (1 <= c <= n-2, so neither neighbour needs a bound check):
// only c+1 and c-1 are examined.  `ambiguous` is raised when the match lies within 2e-4 of the far
// edge of its +-0.5001 band -- the only situation in which a second candidate (c for an upward
// match, c-2 for a downward one) can match as well and relocate_closed() could differ.  The
// margin (0.4998 vs the 0.4999 where the bands start to overlap) is 1e-4, ten orders above the
// rounding of the subtractions.  Callers fall back to relocate_closed() when any lane is ambiguous
// or sits on a face cell.
CBET_HD int relocate_fast_interior(int c, double f, bool &ambiguous)
{
    const double half = 0.5001, safe = 0.4998;
    const double fc = (double)c;
    const double au = fabs((fc + 1.0) - f), ad = fabs((fc - 1.0) - f);
    const bool up = au < half, dn = ad < half;
    ambiguous = ambiguous || (up && au > safe) || (dn && ad > safe);
    return up ? c + 1 : (dn ? c - 1 : c);
}

}  // namespace cbet
#endif
