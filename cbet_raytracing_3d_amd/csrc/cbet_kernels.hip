// cbet_kernels.hip -- gfx950 (CDNA4) kernels of the ray-integrator path.
//
// Built with -ffp-contract=off: every per-ray fp64 operation below is the single IEEE operation the
// reference's statement performs (/root/reference/launch_ray_XZ.cu:117-359), in the same order, so
// a ray's trajectory, absorbed energy and the eight deposit values are the values the CPU oracle
// computes; only the order in which different rays' deposits are summed is free.
//
// Formulation (MI355X-first, not the reference's):
//   * k_tabulate    : the radial (r, ne, Te) profile is evaluated ONCE per node into two node
//                     tables in HBM, ne3d and kappa3d (= ed/ncrit*nuei*dt, launch_ray_XZ.cu:296-305
//                     without the trailing *uray).  The reference re-interpolates the profile eight
//                     times per ray-step (8 bisections + 9 sqrt + 9 div); here a ray-step is seven
//                     8-byte gathers and ~60 flops, no sqrt/div.
//   * k_trace       : one wavefront (64 lanes) = one ray bundle = one 8x8-ray patch of the beam
//                     cross section (patches in Morton order, dead patches dropped, culled rays
//                     are idle lanes), so a wave's gathers and deposits fall in a few
//                     neighbouring cells.
//       DEPOSIT = GLOBAL : 8 global_atomic_add_f64 per ray-step (the reference's scheme).
//       DEPOSIT = LDS    : deposits are combined in a wave-private, toroidally indexed LDS window
//                     (W^3 fp64 accumulators + tags); a slot is written back with one global atomic
//                     when the bundle has moved on and another node claims it, and at wave end.
#include <hip/hip_runtime.h>

#include "cbet_device.h"

namespace cbet {
namespace {

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

// ---------------------------------------------------------------------------------------------
// Node tables.  One thread per node, grid-stride; the 3 x nprofile profile is staged in LDS
// (the one thing kept from the reference's layout, launch_ray_XZ.cu:136-150).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_tabulate(const TabulateArgs a)
{
    extern __shared__ double s_prof[];
    double *s_r = s_prof, *s_ne = s_prof + a.nprofile, *s_te = s_prof + 2 * a.nprofile;
    for (int i = threadIdx.x; i < a.nprofile; i += blockDim.x) {
        s_r[i] = a.r[i];
        s_ne[i] = a.ne[i];
        s_te[i] = a.te[i];
    }
    __syncthreads();
    const long total = (long)a.nx * a.ny * a.nz;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int k = (int)(idx % a.nz);
        const long ij = idx / a.nz;
        const int j = (int)(ij % a.ny);
        const int i = (int)(ij / a.ny);
        // launch_ray_XZ.cu:296 -- node radius, squares summed x,y,z
        const double xc = i * a.dx + a.xmin, yc = j * a.dy + a.ymin, zc = k * a.dz + a.zmin;
        const double rho = sqrt(xc * xc + yc * yc + zc * zc);
        const double ed = interp_table(s_ne, s_r, rho, a.nprofile);     // :297
        const double etemp = interp_table(s_te, s_r, rho, a.nprofile);  // :298
        const double eta = 5.2e-5 * 10.0 / (etemp * sqrt(etemp));       // :299
        const double nuei = (1e6 * ed * (kEc * kEc) / kMe) * eta;       // :300
        a.ne3d[idx] = ed;
        a.kap3d[idx] = ed / a.ncrit * nuei * a.dt;                      // :305 up to "* uray"
    }
}

// ---------------------------------------------------------------------------------------------
// Trace kernel helpers
// ---------------------------------------------------------------------------------------------
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

// launch_ray_XZ.cu:282-292 : nearest-node update whose lower bound follows the index it mutates.
__device__ __forceinline__ int relocate(int c, double f, int n)
{
    const double half = 0.5001;  // :132
    int q = min(n - 1, c + 1);
    while (q >= max(0, c - 1)) {
        c = (fabs(q - f) < half) ? q : c;
        --q;
    }
    return c;
}

__device__ __forceinline__ void global_add(double *p, double v)
{
    // native global_atomic_add_f64, no CAS loop (checked in the ISA)
    unsafeAtomicAdd(p, v);
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// Wave-private LDS write-combining window for the deposits.
//
// slot(i,j,k) = the node's haloed indices taken modulo W per axis (a W^3 torus), tag = the node's
// flat haloed index.  All 64 lanes of the wave run this in lock step (one wave per workgroup, so
// no other wave touches the window); LDS operations of one wave execute in program order.
//   fast path : read the 8 tags; where tag == node, ds_add_f64 the weight.
//   slow path : per corner, retry until done, each round in two ordered phases: (1) re-read the
//               tag, lanes that match add; (2) the rest CAS the tag to their node -- exactly one
//               lane per slot wins, swaps its weight in as the new accumulator value and writes
//               the old (tag, sum) back to HBM with one atomic.
// Invariant: an add under tag T is always issued before the instruction that replaces T, so a
// swapped-out sum holds every add made under the old tag and nothing else.
// ---------------------------------------------------------------------------------------------
template <int WL>
struct LdsWindow {
    static constexpr int W = 1 << WL;
    static constexpr int NSLOT = W * W * W;
    double *val;
    unsigned *tag;

    __device__ __forceinline__ unsigned slot(int i, int j, int k) const
    {
        return (unsigned)((((i & (W - 1)) << WL) | (j & (W - 1))) << WL | (k & (W - 1)));
    }
    __device__ __forceinline__ void clear(int lane)
    {
        for (int s = lane; s < NSLOT; s += kWave) {
            val[s] = 0.0;
            tag[s] = kEmptyTag;
        }
    }
    __device__ __forceinline__ void add(unsigned s, double w)
    {
        __hip_atomic_fetch_add(&val[s], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // Write every live slot back (wave end).
    __device__ __forceinline__ int flush(int lane, double *edep)
    {
        int n = 0;
        for (int s = lane; s < NSLOT; s += kWave) {
            const unsigned t = tag[s];
            if (t != kEmptyTag) {
                global_add(&edep[t], val[s]);
                ++n;
            }
        }
        return n;
    }
};

template <int WL>
__device__ __forceinline__ void lds_deposit8(LdsWindow<WL> &win, bool pending, const unsigned (&slot)[8],
                                             const unsigned (&node)[8], const double (&w)[8],
                                             double *edep, int &n_evict)
{
    unsigned miss = 0;
    if (pending) {
        unsigned t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            t[c] = __hip_atomic_load(&win.tag[slot[c]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (t[c] == node[c])
                win.add(slot[c], w[c]);
            else
                miss |= 1u << c;
        }
    }
    if (!__any(miss != 0)) return;
    __builtin_amdgcn_wave_barrier();  // every fast-path add is issued before any slot changes owner
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        bool pend = (miss >> c) & 1u;
        while (__any(pend)) {
            unsigned t = 0;
            if (pend)
                t = __hip_atomic_load(&win.tag[slot[c]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // phase 1: lanes whose node owns the slot add.  This must complete, for ALL lanes,
            // before phase 2 lets any lane hand the slot to another node -- hence two separate
            // statements with a wave barrier between them, not an if/else whose block order the
            // compiler chooses.
            if (pend && t == node[c]) {
                win.add(slot[c], w[c]);
                pend = false;
            }
            __builtin_amdgcn_wave_barrier();
            // phase 2: the others try to claim the slot; one lane per slot wins the CAS, swaps its
            // weight in as the new sum and writes the previous owner's sum back to HBM.  Losers
            // (and lanes whose node just became the owner) go round again.
            if (pend) {
                unsigned expect = t;
                const bool won = __hip_atomic_compare_exchange_strong(
                    &win.tag[slot[c]], &expect, node[c], __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_WORKGROUP);
                if (won) {
                    const unsigned long long old = __hip_atomic_exchange(
                        reinterpret_cast<unsigned long long *>(&win.val[slot[c]]),
                        (unsigned long long)__double_as_longlong(w[c]), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (t != kEmptyTag) {
                        global_add(&edep[t], __longlong_as_double((long long)old));
                        ++n_evict;
                    }
                    pend = false;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The ray integrator.  DEPOSIT: 1 = global atomics, 2 = LDS window of edge 2^WL.
// ---------------------------------------------------------------------------------------------
template <int DEPOSIT, int WL>
__global__ void __launch_bounds__(kWave) k_trace(const TraceArgs a)
{
    constexpr int NSLOT = (DEPOSIT == 2) ? (1 << (3 * WL)) : 1;
    __shared__ double s_val[NSLOT];
    __shared__ unsigned s_tag[NSLOT];
    const int lane = threadIdx.x;

    LdsWindow<WL> win{s_val, s_tag};
    if (DEPOSIT == 2) {
        win.clear(lane);
        __syncthreads();
    }

    // which bundle: interleaved sharding over (beam, bundle) pairs
    const long g = a.shard_index + (long)a.shard_count * blockIdx.x;
    if (g >= a.total_bundles) return;  // wave-uniform; nothing deposited yet
    const int beam = a.beam_lo + (int)(g / a.bundles_per_beam);
    const int li = (int)(g % a.bundles_per_beam) * kWave + lane;

    Ray s;
    const int pre_raynum = li < a.nlive ? a.live[li] : -1;  // -1: hole in the 8x8 patch
    bool alive = pre_raynum >= 0;
    if (alive) alive = launch_ray(a, beam, pre_raynum, s);
    const int launched = alive ? 1 : 0;

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long sYn = nz, sXn = (long)ny * nz;                 // node-table strides
    const int sYh = nz + 2, sXh = (ny + 2) * (nz + 2);        // haloed edep strides (:5-7)
    int nsteps = 0, n_atomics = 0, n_evict = 0;

    for (int tt = 0; tt < a.nt; ++tt) {                        // :207
        if (!__any(alive)) break;
        unsigned slot[8], node[8];
        double wgt[8];
        if (alive) {
            // :212-238 neighbours, one-sided at the faces
            int im = s.ci - 1, ip = s.ci + 1, jm = s.cj - 1, jp = s.cj + 1, km = s.ck - 1, kp = s.ck + 1;
            if (s.ci == 0) { ip = 2; im = 0; } else if (s.ci == nx - 1) { ip = nx - 1; im = nx - 3; }
            if (s.cj == 0) { jp = 2; jm = 0; } else if (s.cj == ny - 1) { jp = ny - 1; jm = ny - 3; }
            if (s.ck == 0) { kp = 2; km = 0; } else if (s.ck == nz - 1) { kp = nz - 1; km = nz - 3; }
            // :254-265 six gathers from the node table
            const long row = (long)s.ci * sXn + (long)s.cj * sYn;
            const double ne_xp = a.ne3d[(long)ip * sXn + (long)s.cj * sYn + s.ck];
            const double ne_xm = a.ne3d[(long)im * sXn + (long)s.cj * sYn + s.ck];
            const double ne_yp = a.ne3d[(long)s.ci * sXn + (long)jp * sYn + s.ck];
            const double ne_ym = a.ne3d[(long)s.ci * sXn + (long)jm * sYn + s.ck];
            const double ne_zp = a.ne3d[row + kp];
            const double ne_zm = a.ne3d[row + km];
            // :268-273 kick then drift
            s.vx -= a.xconst * (ne_xp - ne_xm);
            s.vy -= a.yconst * (ne_yp - ne_ym);
            s.vz -= a.zconst * (ne_zp - ne_zm);
            s.px += s.vx * a.dt;
            s.py += s.vy * a.dt;
            s.pz += s.vz * a.dt;
            // :276-292
            const double fx = (s.px - a.xmin) * a.inv_dx;
            const double fy = (s.py - a.ymin) * a.inv_dy;
            const double fz = (s.pz - a.zmin) * a.inv_dz;
            s.ci = relocate(s.ci, fx, nx);
            s.cj = relocate(s.cj, fy, ny);
            s.ck = relocate(s.ck, fz, nz);
            // :296-311 absorption at the new node
            double inc;
            if (a.absorption == 1) {
                inc = a.kap3d[(long)s.ci * sXn + (long)s.cj * sYn + s.ck] * s.uray;
                s.uray -= inc;
            } else {
                inc = s.uray;
            }
            // :319-339 weights
            const double ox = fx - s.ci - 0.5, oy = fy - s.cj - 0.5, oz = fz - s.ck - 0.5;
            const double dm = 1.0 - fabs(ox), dn = 1.0 - fabs(oy), dl = 1.0 - fabs(oz);
            const double a1 = (1.0 - dl) * (1.0 - dn) * (1.0 - dm);
            const double a2 = (1.0 - dl) * (1.0 - dn) * dm;
            const double a3 = dl * (1.0 - dn) * (1.0 - dm);
            const double a4 = dl * (1.0 - dn) * dm;
            const double a5 = (1.0 - dl) * dn * (1.0 - dm);
            const double a6 = (1.0 - dl) * dn * dm;
            const double a7 = dl * dn * (1.0 - dm);
            const double a8 = dl * dn * dm;
            const int sx = (ox < 0) ? -1 : 1, sy = (oy < 0) ? -1 : 1, sz = (oz < 0) ? -1 : 1;
            // :341-348 targets, reference order
            const int hi = s.ci + 1, hj = s.cj + 1, hk = s.ck + 1;
            const int base = hi * sXh + hj * sYh + hk;
            wgt[0] = a1 * inc; node[0] = base;
            wgt[1] = a2 * inc; node[1] = base + sx * sXh;
            wgt[2] = a3 * inc; node[2] = base + sz;
            wgt[3] = a4 * inc; node[3] = base + sx * sXh + sz;
            wgt[4] = a5 * inc; node[4] = base + sy * sYh;
            wgt[5] = a6 * inc; node[5] = base + sx * sXh + sy * sYh;
            wgt[6] = a7 * inc; node[6] = base + sy * sYh + sz;
            wgt[7] = a8 * inc; node[7] = base + sx * sXh + sy * sYh + sz;
            if (DEPOSIT == 2) {
                slot[0] = win.slot(hi, hj, hk);
                slot[1] = win.slot(hi + sx, hj, hk);
                slot[2] = win.slot(hi, hj, hk + sz);
                slot[3] = win.slot(hi + sx, hj, hk + sz);
                slot[4] = win.slot(hi, hj + sy, hk);
                slot[5] = win.slot(hi + sx, hj + sy, hk);
                slot[6] = win.slot(hi, hj + sy, hk + sz);
                slot[7] = win.slot(hi + sx, hj + sy, hk + sz);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) global_add(&a.edep[node[c]], wgt[c]);
                n_atomics += 8;
            }
            ++nsteps;
        }
        if (DEPOSIT == 2) lds_deposit8<WL>(win, alive, slot, node, wgt, a.edep, n_evict);
        // :351-356
        if (alive && (s.uray <= s.ustop || s.px < a.xlo || s.px > a.xhi || s.py < a.ylo ||
                      s.py > a.yhi || s.pz < a.zlo || s.pz > a.zhi))
            alive = false;
    }

    if (DEPOSIT == 2) {
        __syncthreads();
        n_atomics += win.flush(lane, a.edep) + n_evict;
    }
    // counters: one atomic per wave and counter
    const int tot_steps = wave_sum(nsteps), tot_rays = wave_sum(launched), tot_at = wave_sum(n_atomics),
              tot_ev = wave_sum(n_evict);
    if (lane == 0) {
        atomicAdd(&a.counters[kCntSteps], (unsigned long long)tot_steps);
        atomicAdd(&a.counters[kCntRays], (unsigned long long)tot_rays);
        atomicAdd(&a.counters[kCntGlobalAtomics], (unsigned long long)tot_at);
        atomicAdd(&a.counters[kCntEvictions], (unsigned long long)tot_ev);
    }
}

}  // namespace

hipError_t launch_tabulate(const TabulateArgs &a, hipStream_t stream)
{
    const long total = (long)a.nx * a.ny * a.nz;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // 256 CUs x 16 blocks, grid-stride the rest
    const size_t lds = sizeof(double) * 3 * (size_t)a.nprofile;
    hipLaunchKernelGGL(k_tabulate, dim3((unsigned)blocks), dim3(256), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_trace(const TraceArgs &a, int variant, int window_log2, hipStream_t stream)
{
    const long waves = (a.total_bundles - a.shard_index + a.shard_count - 1) / a.shard_count;
    if (waves <= 0) return hipSuccess;
    const dim3 grid((unsigned)waves), block(kWave);
    if (variant == CBET_KERNEL_GLOBAL_ATOMICS) {
        hipLaunchKernelGGL((k_trace<1, 1>), grid, block, 0, stream, a);
    } else if (window_log2 == 4) {
        hipLaunchKernelGGL((k_trace<2, 4>), grid, block, 0, stream, a);
    } else {
        hipLaunchKernelGGL((k_trace<2, 3>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace cbet
