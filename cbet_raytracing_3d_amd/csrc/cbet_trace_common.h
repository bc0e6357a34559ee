// cbet_trace_common.h -- device helpers shared by the ray-integrator kernels (cbet_kernels.hip: the two
// cross-check formulations; cbet_trace_window.hip: the shipped one).  Citations are into /root/reference/.
// Everything here is compiled with -ffp-contract=off: one IEEE operation per reference statement.
#ifndef CBET_TRACE_COMMON_H_
#define CBET_TRACE_COMMON_H_

#include <hip/hip_runtime.h>

#include "cbet_device.h"
#include "cbet_relocate.h"

namespace cbet {

// ---------------------------------------------------------------------------------------------
// launch_ray_XZ.cu:16-63 -- clamped piecewise-linear lookup, bisection; both abscissa orders.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double interp_table(const double *y, const double *x, const double xp, int n)
{
    unsigned lo, hi, mid;
    if (x[0] <= x[n - 1]) {
        if (xp <= x[0]) return y[0];
        if (xp >= x[n - 1]) return y[n - 1];
        lo = 0;
        hi = n - 1;
        mid = (lo + hi) >> 1;
        while (lo < hi - 1) {
            if (x[mid] >= xp) hi = mid; else lo = mid;
            mid = (lo + hi) >> 1;
        }
    } else {
        if (xp >= x[0]) return y[0];
        if (xp <= x[n - 1]) return y[n - 1];
        lo = 0;
        hi = n - 1;
        mid = (lo + hi) >> 1;
        while (lo < hi - 1) {
            if (x[mid] <= xp) lo = mid; else hi = mid;
            mid = (lo + hi) >> 1;
        }
    }
    return y[mid] + (y[mid + 1] - y[mid]) / (x[mid + 1] - x[mid]) * (xp - x[mid]);
}

struct Ray {
    double px, py, pz, vx, vy, vz, uray, ustop;
    int ci, cj, ck;
};

// First node q in [0,n) with |q*d+lo - p| <= tol, else 0 (launch_ray_XZ.cu:162-180).  Only nodes
// next to p can satisfy the predicate, so the upward scan is restricted to a 5-node window; the
// predicate itself is the reference's.
__device__ __forceinline__ int first_node_within(double p, double lo, double d, double tol, int n)
{
    double f = (p - lo) / d;
    int g = (f > -4.0 && f < (double)n + 4.0) ? (int)floor(f) : -8;
    int found = 0;
    bool have = false;
    for (int q = g - 2; q <= g + 2; ++q) {
        if (q < 0 || q >= n || have) continue;
        if (fabs(q * d + lo - p) <= tol) {
            found = q;
            have = true;
        }
    }
    return found;
}

// launch_ray_XZ.cu:65-115 + :162-204 : launch point, power, first cell, launch wave-vector.
__device__ __forceinline__ bool launch_ray(const TraceArgs &a, int beam, int pre_raynum, Ray &s)
{
    const int rpz = a.rpz, rpz2 = rpz * rpz;
    const int tile = pre_raynum / rpz2, within = pre_raynum % rpz2;   // :70-71
    const int ry = tile / a.zones * rpz + within / rpz;               // :72
    const int rx = tile % a.zones * rpz + within % rpz;               // :73
    // :76-92 the repeated-addition loops are tabulated on the host (same additions, same order)
    double x0 = a.xlaunch[rx];
    double y0 = a.ylaunch[ry];
    const double ref = sqrt(x0 * x0 + y0 * y0);                       // :94
    double z0 = a.z_launch;                                           // :97

    const double bnx = a.beam_norm[beam * 3 + 0], bny = a.beam_norm[beam * 3 + 1],
                 bnz = a.beam_norm[beam * 3 + 2];
    double c1, s1, c2, s2;
    if (a.bbeam_norm) {  // main.cu:121-129 host trig, 4 per beam
        c1 = a.bbeam_norm[4 * beam + 0];
        s1 = a.bbeam_norm[4 * beam + 1];
        c2 = a.bbeam_norm[4 * beam + 2];
        s2 = a.bbeam_norm[4 * beam + 3];
    } else {             // :99-100 on the device
        const double theta1 = acos(bnz);
        const double theta2 = atan2(bny * kFocal, kFocal * bnx);
        c1 = cos(theta1);
        s1 = sin(theta1);
        c2 = cos(theta2);
        s2 = sin(theta2);
    }
    const double keep = x0;                                           // :102-111
    x0 = x0 * c1 + z0 * s1;
    z0 = z0 * c1 - keep * s1;
    const double keep2 = x0;
    x0 = x0 * c2 - y0 * s2;
    y0 = y0 * c2 + keep2 * s2;

    s.px = x0;
    s.py = y0;
    s.pz = z0;
    s.uray = a.uray_mult * interp_table(a.pow_r, a.phase_r, ref, CBET_NPHASE);  // :113
    s.ustop = 0.05 * s.uray;                                                    // :351
    if (!(ref <= kBeamMax)) return false;                                       // :114

    s.ci = first_node_within(s.px, a.xmin, a.dx, a.tol_x, a.nx);      // :162-180
    s.cj = first_node_within(s.py, a.ymin, a.dy, a.tol_y, a.ny);
    s.ck = first_node_within(s.pz, a.zmin, a.dz, a.tol_z, a.nz);

    // :186-204 ne at the launch node == the tabulated node value
    const double ne0 = a.ne3d[((long)s.ci * a.ny + s.cj) * a.nz + s.ck];
    const double w = sqrt((a.omega * a.omega - ne0 * 1e6 * (kEc * kEc) / ((double)kMe * kE0)) / (kC * kC));
    double vx = -1 * bnx, vy = -1 * bny, vz = -1 * bnz;
    const double knorm = sqrt(vx * vx + vy * vy + vz * vz);
    s.vx = (kC * kC) * ((vx / knorm) * w) / a.omega;
    s.vy = (kC * kC) * ((vy / knorm) * w) / a.omega;
    s.vz = (kC * kC) * ((vz / knorm) * w) / a.omega;
    return true;
}

// Bounds-audited build (tests/test_gpu_bounds_audit.py, -DCBET_DEBUG_BOUNDS): every grid atomic, node-table
// gather and LDS accumulate is range-checked against the limits the launch put into TraceArgs; a violation
// is counted and the access skipped.  Never shipped.
#ifdef CBET_DEBUG_BOUNDS
__device__ __forceinline__ bool audit_fail(const TraceArgs &a)
{
    atomicAdd(a.audit_count, 1ull);
    return true;
}
#define CBET_AUDIT(a, cond) ((cond) || !audit_fail(a))
#else
#define CBET_AUDIT(a, cond) true
#endif

__device__ __forceinline__ void global_add(const TraceArgs &a, double *p, double v)
{
#ifdef CBET_DEBUG_BOUNDS
    if (!(p >= a.audit_lo && p < a.audit_hi)) { audit_fail(a); return; }
#else
    (void)a;
#endif
    // native global_atomic_add_f64, no CAS loop (checked in the ISA)
    unsafeAtomicAdd(p, v);
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// 8-byte gather from a node table by 32-bit element index: uniform base + zero-extended 32-bit
// byte offset, which the backend turns into the saddr+voffset form of global_load_dwordx2 (no
// 64-bit address arithmetic per lane).  Valid while 8*nodes < 2^32 (checked on the host).
template <bool IDX64>
__device__ __forceinline__ double node_load(const TraceArgs &a, const double *base, unsigned idx)
{
#ifdef CBET_DEBUG_BOUNDS
    if (!(idx < a.audit_nodes)) { audit_fail(a); return 0.0; }
#else
    (void)a;
#endif
    if (IDX64) return base[idx];
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (idx * 8u));
}

// The same gather from a beam's haloed gain grid (CBET hooks); only the audited range differs.
template <bool IDX64>
__device__ __forceinline__ double gain_load(const TraceArgs &a, const double *base, unsigned idx)
{
#ifdef CBET_DEBUG_BOUNDS
    if (!(idx < a.audit_hsize)) { audit_fail(a); return 0.0; }
#else
    (void)a;
#endif
    if (IDX64) return base[idx];
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (idx * 8u));
}

// Two z-adjacent entries (idx, idx + 1) of the gain grid in ONE 16-byte gather (8-byte aligned: the hardware takes it).
typedef double gain_pair_t __attribute__((ext_vector_type(2), aligned(8)));
template <bool IDX64>
__device__ __forceinline__ gain_pair_t gain_load2(const TraceArgs &a, const double *base, unsigned idx)
{
#ifdef CBET_DEBUG_BOUNDS
    if (!(idx + 1u < a.audit_hsize)) { audit_fail(a); return gain_pair_t{0.0, 0.0}; }
#else
    (void)a;
#endif
    if (IDX64) return *reinterpret_cast<const gain_pair_t *>(base + idx);
    return *reinterpret_cast<const gain_pair_t *>(reinterpret_cast<const char *>(base) + (idx * 8u));
}

// phi(x) = (exp(x) - 1) / x, |x| <= 1: degree-17 Horner polynomial of plain multiplies and adds, the
// operation sequence the CPU checker of the CBET stage evaluates.  CBET extension only.
__device__ __forceinline__ double phi_det(double x)
{
    double p = 1.0 / 6402373705728000.0;
    p = p * x + 1.0 / 355687428096000.0;
    p = p * x + 1.0 / 20922789888000.0;
    p = p * x + 1.0 / 1307674368000.0;
    p = p * x + 1.0 / 87178291200.0;
    p = p * x + 1.0 / 6227020800.0;
    p = p * x + 1.0 / 479001600.0;
    p = p * x + 1.0 / 39916800.0;
    p = p * x + 1.0 / 3628800.0;
    p = p * x + 1.0 / 362880.0;
    p = p * x + 1.0 / 40320.0;
    p = p * x + 1.0 / 5040.0;
    p = p * x + 1.0 / 720.0;
    p = p * x + 1.0 / 120.0;
    p = p * x + 1.0 / 24.0;
    p = p * x + 1.0 / 6.0;
    p = p * x + 0.5;
    p = p * x + 1.0;
    return p;
}

// ... and for |x| < 2^-5 the series cut after x^7: the first dropped term, x^8 / 9!, is below 2.6e-18 of phi there, far
// inside the rounding of the sum.  The trace kernel takes this form when EVERY live lane of the wave is that small (one
// ballot) -- most wave-steps: K ds is a few 1e-3 outside the hot spots of the exchange -- which takes 20 of the 34
// dependent fp64 operations out of the step's critical path.  The CPU checker always evaluates the long form; the two
// agree to the last bit or two, and the kernel is held to the checker at 1e-9.
__device__ __forceinline__ double phi_small(double x)
{
    // (fused multiply-adds: 7 operations for 14; the checker's unfused Horner form differs in the last bit or two)
    double p = 1.0 / 40320.0;
    p = __builtin_fma(p, x, 1.0 / 5040.0);
    p = __builtin_fma(p, x, 1.0 / 720.0);
    p = __builtin_fma(p, x, 1.0 / 120.0);
    p = __builtin_fma(p, x, 1.0 / 24.0);
    p = __builtin_fma(p, x, 1.0 / 6.0);
    p = __builtin_fma(p, x, 0.5);
    p = __builtin_fma(p, x, 1.0);
    return p;
}

// sqrt of a positive normal number far from the ends of the exponent range (a squared speed, cm^2/s^2), within an ulp:
// reciprocal-square-root estimate, one Goldschmidt step, two corrections -- without the range scaling and the special
// cases of the library sqrt (13 instructions for 27).  CBET extension only (the path length of a step); the reference
// path has no square root in its step.
__device__ __forceinline__ double sqrt_speed(double v2)
{
    const double r0 = __builtin_amdgcn_rsq(v2);
    double g = v2 * r0, h = 0.5 * r0;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
    g = __builtin_fma(__builtin_fma(-g, g, v2), h, g);
    return __builtin_fma(__builtin_fma(-g, g, v2), h, g);
}

// The work item of workgroup `w` of a launch: which (beam, patch) bundle.  The list is beam-major -- consecutive
// workgroups are neighbouring patches of one beam and share table lines in L2/MALL -- with each beam's patches
// longest rays first (a globally longest-first order and a patch-major order were measured 9-17 % slower).  A
// shard is a CONTIGUOUS 1/shard_count of that list (first_item .. first_item + count): a rank then walks ~7.5
// whole beams of the 60 and touches only their stretch of the 537 MB record table, where an interleaved split
// makes every rank touch every beam's (one rank's 1/8 share: 3.2 ms contiguous, 3.4 ms interleaved).
// (Round 3 measured the alternative of visiting a share in LENGTH CLASSES -- every beam's longest bundles first,
// beam by beam inside a class, so that a 1/8 share ends on short bundles: the launch's tail shrinks, but ~9 beams'
// stretches of the record table are then live at once instead of ~2 and the whole pass is 15-22 % slower; a 1/8
// share alone 3.45 ms against 3.32.  profiles/r3/experiments/length_class_order.log.)
__device__ __forceinline__ bool work_item(const TraceArgs &a, long w, int &beam, int &patch)
{
    const long g = a.first_item + w;
    if (w >= a.item_count) return false;  // wave-uniform
    const int beam_local = (int)(g / a.bundles_per_beam);
    patch = (int)(g - (long)beam_local * a.bundles_per_beam);
    beam = a.beam_lo + beam_local;
    return true;
}

}  // namespace cbet
#endif
