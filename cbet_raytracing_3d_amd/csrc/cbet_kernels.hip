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
//   * k_trace       : one wavefront (64 lanes, one workgroup) = one ray bundle = one 8x8-ray patch
//                     of the beam cross section (the host orders the patches, drops dead ones and
//                     marks culled rays as holes), so a wave's gathers and deposits fall in a few
//                     neighbouring cells.  The step loop is software-pipelined (next step's gathers
//                     are issued right after relocation).  Three deposit schemes, template DEPOSIT:
//       1  GLOBAL : 8 global_atomic_add_f64 per ray-step (the reference's scheme, kept as baseline).
//       2  TAGGED : wave-private toroidal LDS tile with node tags; slots are claimed by LDS CAS and
//                   written back with one global atomic when another node claims them.
//       3  WINDOW : (default) wave-private dense LDS tiles without tags whose origins follow the
//                   bundle; a slab leaving a box is flushed with one global atomic per node.  See
//                   MovingWindow and DESIGN.md 4.2 for the measurements behind each choice.
#include <hip/hip_runtime.h>

#include "cbet_device.h"
#include "cbet_relocate.h"

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

// Two tables over ONE abscissa (ne and Te share r_data, launch_ray_XZ.cu:297-298): the bisection
// depends only on (x, xp), so it is done once and both values are interpolated from the same
// segment -- bit for bit what two interp_table() calls return.
__device__ __forceinline__ void interp_table2(const double *y1, const double *y2, const double *x, const double xp,
                                              int n, double &o1, double &o2)
{
    const bool ascending = x[0] <= x[n - 1];
    if (ascending ? (xp <= x[0]) : (xp >= x[0])) { o1 = y1[0]; o2 = y2[0]; return; }
    if (ascending ? (xp >= x[n - 1]) : (xp <= x[n - 1])) { o1 = y1[n - 1]; o2 = y2[n - 1]; return; }
    unsigned lo = 0, hi = n - 1, mid = (lo + hi) >> 1;
    while (lo < hi - 1) {
        const bool go_low = ascending ? (x[mid] >= xp) : !(x[mid] <= xp);  // :31 / :52 (as written there)
        if (go_low) hi = mid; else lo = mid;
        mid = (lo + hi) >> 1;
    }
    const double dx = x[mid + 1] - x[mid], t = xp - x[mid];
    o1 = y1[mid] + (y1[mid + 1] - y1[mid]) / dx * t;
    o2 = y2[mid] + (y2[mid + 1] - y2[mid]) / dx * t;
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
        double ed, etemp;                                               // :297-298
        interp_table2(s_ne, s_te, s_r, rho, a.nprofile, ed, etemp);
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

#ifdef CBET_DEBUG_BOUNDS
// Bounds-audited build (tests/test_gpu_bounds_audit.py): every grid atomic, node-table gather and
// LDS accumulate is range-checked; a violation is counted and the access skipped.  Never shipped.
__device__ const double *g_audit_edep_lo, *g_audit_edep_hi;
__device__ unsigned long long g_audit_nodes;
__device__ unsigned long long g_audit_hsize;   // CBET hooks: entries of a beam's haloed gain grid
__device__ unsigned long long g_audit_violations;
__device__ __forceinline__ bool audit_fail() { atomicAdd(&g_audit_violations, 1ull); return true; }
#define CBET_AUDIT(cond) ((cond) || !audit_fail())
#else
#define CBET_AUDIT(cond) true
#endif

__device__ __forceinline__ void global_add(double *p, double v)
{
#ifdef CBET_DEBUG_BOUNDS
    if (!(p >= g_audit_edep_lo && p < g_audit_edep_hi)) { audit_fail(); return; }
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
        if (CBET_AUDIT(s < (unsigned)NSLOT))
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
// Dense moving window (DEPOSIT = 3): the tuned deposit scheme.
//
// A wave-private W^3 tile of fp64 accumulators, addressed toroidally (haloed node index mod W per
// axis) and covering the box [o, o+W) per axis, where the wave-uniform origin o follows the
// bundle: each step the cell of a proxy ray (the patch's middle lane while it is alive) is kept
// inside the box's central band.  Moving the origin by one cell along an axis retires one W x W
// slab -- for W = 8 exactly one accumulator per lane: read it, add it to HBM with one atomic if
// non-zero, zero it.  No tags, no compare-and-swap: a lane whose eight target nodes lie in the
// box issues eight unconditional ds_add_f64; a lane that strays outside (a stretched bundle)
// deposits straight to HBM for that step.  Everything is lock-step within one wave (one wave per
// workgroup), so the origin, the shifts and the slab loops are scalar.
// ---------------------------------------------------------------------------------------------
template <int WL, int RL, int NC = 1>
struct MovingWindow {
    static constexpr int W = 1 << WL;
    static constexpr int R = 1 << RL;     // privatised copies of the tile, selected by lane & (R-1)
    static constexpr int S = W - 2;       // largest valid offset of a lane's low corner
    // Padded layout (in doubles).  ds_add_f64 costs the CU ~8 cycles when the lanes' addresses fall on
    // different bank pairs, +2 per lane sharing a bank, +3 per lane sharing an address (measured,
    // scripts/ubench/lds_atomic.hip).  A bundle's footprint is a few nodes wide per axis, so rows are
    // padded to W+1 and planes to W*(W+1)+4: neighbours in y land 9 bank pairs apart, neighbours in x
    // 12 apart (of 16), and the copies of one node 8 apart.
    static constexpr int YS = W + 1;
    static constexpr int XS = W * YS + 4;
    static constexpr int CS = W * XS + 8;          // copy stride
    static constexpr int NDOUBLES = R * CS;
    double *val;                          // NDOUBLES accumulators
    int ox, oy, oz;
    // NC > 1 (CBET field pass): NC - 1 further, unpadded W^3 tiles at val + coff + (q - 1) * DT, q = 1..,
    // flushed to grids `gstride` doubles apart in HBM (components 1.. are never deferred)
    static constexpr int DT = W * W * W;
    int limit;                            // doubles addressable from val (all boxes' and components' tiles)
    int coff;
    long gstride;
    static __device__ __forceinline__ int addr_d(int rx, int ry, int rz) { return (rx * W + ry) * W + rz; }

    static __device__ __forceinline__ int addr(int rx, int ry, int rz) { return rx * XS + ry * YS + rz; }

    __device__ __forceinline__ void init(int lane, int hx, int hy, int hz)
    {
        for (int s = lane; s < NDOUBLES; s += kWave) val[s] = 0.0;
        ox = hx - W / 2;
        oy = hy - W / 2;
        oz = hz - W / 2;
    }
    // absolute coordinate in [o, o+W) whose residue mod W is r
    static __device__ __forceinline__ int absolute(int o, int r) { return o + ((r - o) & (W - 1)); }

    // Retire the slab `coord` (absolute, inside the box) of axis AX: take the non-zero sums out of
    // the tile and zero them.  For W = 8 a slab is exactly one accumulator per lane and the sum is
    // handed back in (dv, dn) = (value, flat haloed node index) so that the caller can issue the
    // global atomic LATER, behind the next step's gathers (loads, stores and atomics share one
    // in-order vmcnt on CDNA: an atomic issued before a load delays that load's data by the
    // atomic's ~3000-cycle round trip).  DEFER = false (W = 16, final flush): atomics issued here.
    template <int AX, bool DEFER>
    __device__ __forceinline__ void retire(int coord, int lane, double *edep, int sXh, int sYh, int &n_at,
                                           double &dv, int &dn)
    {
        const int fixed = coord & (W - 1);
#pragma unroll
        for (int e = lane; e < W * W; e += kWave) {
            const int r0 = e >> WL, r1 = e & (W - 1);
            int i, j, k, slot;
            int slot_d = 0;
            if (AX == 0) { i = coord; j = absolute(oy, r0); k = absolute(oz, r1); slot = addr(fixed, r0, r1); slot_d = addr_d(fixed, r0, r1); }
            else if (AX == 1) { i = absolute(ox, r0); j = coord; k = absolute(oz, r1); slot = addr(r0, fixed, r1); slot_d = addr_d(r0, fixed, r1); }
            else { i = absolute(ox, r0); j = absolute(oy, r1); k = coord; slot = addr(r0, r1, fixed); slot_d = addr_d(r0, r1, fixed); }
            if (!CBET_AUDIT((unsigned)((R - 1) * CS + slot) < (unsigned)NDOUBLES)) continue;
            double v = val[slot];
#pragma unroll
            for (int c = 1; c < R; ++c) v += val[c * CS + slot];
            const int node = i * sXh + j * sYh + k;
            if (NC > 1) {
#pragma unroll
                for (int q = 1; q < NC; ++q) {
                    const double vq = val[coff + (q - 1) * DT + slot_d];
                    if (vq != 0.0) {
#ifndef CBET_EXPERIMENT_DROP_FLUSH_ATOMICS
                        global_add(&edep[q * gstride + node], vq);
#endif
                        val[coff + (q - 1) * DT + slot_d] = 0.0;
                        ++n_at;
                    }
                }
            }
            if (v != 0.0) {  // only nodes that received deposits are non-zero, hence valid
                if (DEFER && W * W == kWave) {
                    dv = v;
                    dn = node;
                } else {
#ifndef CBET_EXPERIMENT_DROP_FLUSH_ATOMICS  // timing-only experiment builds (scripts/experiment_*.sh); never shipped
                    global_add(&edep[node], v);
#endif
                }
#pragma unroll
                for (int c = 0; c < R; ++c) val[c * CS + slot] = 0.0;
                ++n_at;
            }
        }
    }
    static __device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

    template <int AX, bool DEFER>
    __device__ __forceinline__ void follow_axis(int &o, bool alive, int lo_corner, int lane, double *edep,
                                                int sXh, int sYh, int &n_at, int &n_wide, unsigned &n_slabs16,
                                                double &dv, int &dn)
    {
        // dead lanes get a neutral offset (mid-box), so no ballot needs the alive mask (masking the
        // ballots on the scalar unit instead was measured slower: scalar-register pressure)
        const int rel = alive ? lo_corner - o : W / 2 - 1;
        // one ballot decides whether anything can happen: a lane on an edge cell or outside
        if (!any_lane((unsigned)(rel - 1) >= (unsigned)(S - 1))) return;
        const bool below = any_lane(rel < 0), at_lo = any_lane(rel <= 0), near_lo = any_lane(rel <= 1);
        const bool above = any_lane(rel > S), at_hi = any_lane(rel >= S), near_hi = any_lane(rel >= S - 1);
        const bool want_down = below || (at_lo && !near_hi);
        const bool want_up = above || (at_hi && !near_lo);
        if (below && above) ++n_wide;  // wave-uniform: the bundle does not fit the box on this axis
        if (want_down && !at_hi) {
            retire<AX, DEFER>(o + W - 1, lane, edep, sXh, sYh, n_at, dv, dn);
            o -= 1;
            n_slabs16 += 1u << 16;  // packed: slabs retired in the high half
        } else if (want_up && !at_lo) {
            retire<AX, DEFER>(o, lane, edep, sXh, sYh, n_at, dv, dn);
            o += 1;
            n_slabs16 += 1u << 16;  // packed: slabs retired in the high half
        }
    }
    // are the lane's 8 targets (low corner lx,ly,lz and its +1 neighbours) inside the box?
    __device__ __forceinline__ bool holds(int lx, int ly, int lz) const
    {
        return (unsigned)(lx - ox) <= (unsigned)S && (unsigned)(ly - oy) <= (unsigned)S &&
               (unsigned)(lz - oz) <= (unsigned)S;
    }
    __device__ __forceinline__ void flush_all(int lane, double *edep, int sXh, int sYh, int &n_at)
    {
        double dv = 0.0;
        int dn = 0;
        for (int t = 0; t < W; ++t) retire<0, false>(ox + t, lane, edep, sXh, sYh, n_at, dv, dn);
    }
    __device__ __forceinline__ void add(int slot, double w)
    {
        if (CBET_AUDIT((unsigned)slot < (unsigned)limit))
            __hip_atomic_fetch_add(&val[slot], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

// Cross-lane moves inside a quad (4 consecutive lanes = 4 rays a quarter cell apart along the
// patch's x axis) through DPP: no LDS, one VALU op per 32-bit half.
template <int CTRL>
__device__ __forceinline__ int quad_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double quad_d(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = quad_i<CTRL>((int)b), hi = quad_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// One level of the in-register pre-reduction: lanes l and l^M whose rays target the same 8 nodes
// (equal key) merge -- the lower lane takes the sum, the upper lane stops depositing (its key
// becomes a unique negative value, so it can never match again).
template <int CTRL, int M>
__device__ __forceinline__ void merge_level(int lane, int &key, double (&w)[8])
{
    const int pk = quad_i<CTRL>(key);
    const bool same = (pk == key) && key >= 0;
    const bool lower = (lane & M) == 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double pw = quad_d<CTRL>(w[c]);
        if (same && lower) w[c] += pw;
    }
    if (same && !lower) key = -2 - lane;
}

// 8-byte gather from a node table by 32-bit element index: uniform base + zero-extended 32-bit
// byte offset, which the backend turns into the saddr+voffset form of global_load_dwordx2 (no
// 64-bit address arithmetic per lane).  Valid while 8*nodes < 2^32 (checked on the host).
template <bool IDX64>
__device__ __forceinline__ double node_load(const double *base, unsigned idx)
{
#ifdef CBET_DEBUG_BOUNDS
    if (!(idx < g_audit_nodes)) { audit_fail(); return 0.0; }
#endif
    if (IDX64) return base[idx];
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (idx * 8u));
}

// The same gather from a beam's haloed gain grid (CBET hooks); only the audited range differs.
template <bool IDX64>
__device__ __forceinline__ double gain_load(const double *base, unsigned idx)
{
#ifdef CBET_DEBUG_BOUNDS
    if (!(idx < g_audit_hsize)) { audit_fail(); return 0.0; }
#endif
    if (IDX64) return base[idx];
    return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (idx * 8u));
}

// ---------------------------------------------------------------------------------------------
// The ray integrator.
//   DEPOSIT 1: 8 global atomics per step   2: tagged LDS window   3: dense moving LDS window
//   WL       : log2 of the LDS window edge (DEPOSIT 2, 3)
//   RL       : log2 of the number of privatised window copies (DEPOSIT 3)
//   PRE      : levels of in-register pre-reduction across neighbouring lanes (DEPOSIT 3; 0, 1, 2)
//   FLIP     : lane-dependent corner order (DEPOSIT 3; see the weights section)
//   TWOBOX   : a second window adopts the lanes that leave the first (DEPOSIT 3, RL = 0, PRE = 0)
//   IDX64    : node tables of >= 2^32 bytes (n > 812)
//   ABSORB   : def.cuh:118 absorption == 1 (false: bookkeeping mode, launch_ray_XZ.cu:307-311)
// ---------------------------------------------------------------------------------------------
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

// CBET != 0 adds the cross-beam-energy-transfer hooks (no reference counterpart): the
// gain coefficient gathered from the eight deposit nodes, ray energy x exp(K ds), and
//   CBET = 1: the deposit is the absorbed energy, as in the reference path;
//   CBET = 4: the field pass -- FOUR grids per beam in one trace: energy x path length with the eight
//             deposit weights (component 0) and energy x displacement x/y/z at the ray's own node
//             (components 1..3).  Components 1..3 have an unpadded 8^3 LDS tile each in box A only
//             (22.2 KB per wave, 7 waves per CU; padded tiles for both boxes left 4 waves per CU and
//             65 ms without any atomic, against 22 ms for the plain pass); lanes homed in box B add
//             their three values straight to HBM.
template <int DEPOSIT, int WL, int RL, int PRE, bool FLIP, bool TWOBOX, bool IDX64, bool ABSORB, int CBET = 0>
__global__ void __launch_bounds__(kWave) k_trace(const TraceArgs a)
{
    constexpr int NC = (CBET == 4) ? 4 : 1;
    constexpr int NSLOT = (DEPOSIT == 3) ? (TWOBOX ? 2 : 1) * MovingWindow<WL, RL>::NDOUBLES
                                         : (DEPOSIT == 2 ? (1 << (3 * WL)) : 1);
    constexpr int NTAG = (DEPOSIT == 2) ? NSLOT : 1;
    constexpr int W = 1 << WL;
    constexpr int NLDS = NSLOT + (NC - 1) * MovingWindow<WL, RL>::DT;   // + components 1.. of box A
    __shared__ double s_val[NLDS];
    __shared__ unsigned s_tag[NTAG];
    const int lane = threadIdx.x;
#ifdef CBET_EXPERIMENT_EXTRA_LDS  // occupancy-sensitivity experiment builds only (scripts/experiment_occupancy.sh)
    __shared__ double s_pad[CBET_EXPERIMENT_EXTRA_LDS / 8];
    if (a.nt < 0) s_pad[lane] = 1.0;  // keep the allocation alive
#endif

#ifdef CBET_EXPERIMENT_TIMELINE  // diagnostic builds only (scripts/experiment_timeline.sh): wave start/end stamps
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    // which bundle: interleaved sharding over (beam, bundle) pairs.  Workgroups are dealt round-robin
    // over the 8 XCDs (b and b+8 share an L2); with xcd_chunk > 0 workgroup b takes work item
    // (b % 8) * xcd_chunk + b / 8, so that consecutive bundles -- neighbouring patches of one beam,
    // which gather the same node-table lines -- run on the same XCD.  Placement only affects speed.
    long w = blockIdx.x;
    if (a.xcd_chunk > 0) {
        w = (long)(blockIdx.x & 7) * a.xcd_chunk + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= a.xcd_chunk) return;
    }
    const long g = a.shard_index + (long)a.shard_count * w;
    if (g >= a.total_bundles) return;  // wave-uniform
    // (beam, patch) of work item g.  Default (phases = 1): beam-major -- consecutive waves are
    // neighbouring patches of one beam and share node-table lines in L2/MALL.  With phases > 1,
    // phase p covers patches [p * phase_len, (p+1) * phase_len) of EVERY beam, beam by beam: a
    // globally longest-first order that ends the launch on short bundles (the drain of a short launch
    // is ~1.2 ms of waiting for the last beam's long bundles, scripts/experiment_timeline.py) -- but it
    // was measured slower overall (cbet_params.order_phases), as was pure patch-major order (-17 %).
    int beam_local, patch;
    {
        const long phase_items = (long)a.nbeams_local * a.phase_len;   // items in every phase but the last
        const int ph = (int)min((long)(a.phases - 1), g / phase_items);
        const long rem = g - ph * phase_items;
        const int len = (ph < a.phases - 1) ? a.phase_len : a.bundles_per_beam - (a.phases - 1) * a.phase_len;
        beam_local = (int)(rem / len);
        patch = ph * a.phase_len + (int)(rem % len);
    }
    const int beam = a.beam_lo + beam_local;
    // beam-resolved deposition (cbet_params.per_beam_grids): beam b accumulates into its own grid,
    // edep[b * grid_stride ...]; otherwise every beam adds into the one grid (grid_stride = 0)
    double *const edep = a.edep + (long)beam * a.grid_stride;
    const int li = patch * kWave + lane;

    Ray s;
    const int pre_raynum = li < a.nlive ? a.live[li] : -1;  // -1: hole in the 8x8 patch
    bool alive = pre_raynum >= 0;
    if (alive) alive = launch_ray(a, beam, pre_raynum, s);
    const int launched = alive ? 1 : 0;

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const int sY = nz, sX = ny * nz;                      // node-table strides (elements)
    const int sYh = nz + 2, sXh = (ny + 2) * (nz + 2);    // haloed edep strides (:5-7)
    unsigned cell = alive ? (unsigned)((s.ci * ny + s.cj) * nz + s.ck) : 0u;
    int nsteps = 0, n_atomics = 0, n_evict = 0;
    // wave-uniform diagnostics, packed two to a scalar register (each < 2^16: nt <= 4 n)
    unsigned w_steps_miss = 0;   // wave-steps << 16 | wave-steps with a window miss
    unsigned w_slabs_wide = 0;   // slabs retired << 16 | wave-steps "too wide" (two boxes: box B live)

    LdsWindow<WL> tagged{s_val, s_tag};
    MovingWindow<WL, RL, NC> win{s_val, 0, 0, 0, NLDS, NSLOT, a.comp_stride};
    // Second box (TWOBOX): after the turning point a bundle fans out to 6-11 cells (scripts/
    // bundle_spread.py), wider than one 8-cell box.  Lanes that fall out of box A are adopted by
    // box B (sticky per-lane home bit); B is created around the first such lane and flushed when
    // its last lane leaves or dies.
    MovingWindow<WL, RL> winB{s_val + (TWOBOX ? MovingWindow<WL, RL>::NDOUBLES : 0), 0, 0, 0,
                              MovingWindow<WL, RL>::NDOUBLES, 0, 0};
    bool homeB = false;     // per lane
    bool b_active = false;  // wave-uniform

    if (DEPOSIT == 2) {
        tagged.clear(lane);
        __syncthreads();
    }
    if (DEPOSIT == 3) {
        const unsigned long long m = __ballot(alive);
        if (m == 0) return;  // whole bundle culled (cannot happen for a listed patch; cheap guard)
        const int src = ((m >> 27) & 1ull) ? 27 : (__ffsll((long long)m) - 1);
        win.init(lane, __builtin_amdgcn_readlane(s.ci, src) + 1, __builtin_amdgcn_readlane(s.cj, src) + 1,
                 __builtin_amdgcn_readlane(s.ck, src) + 1);
        if (TWOBOX) winB.init(lane, 0, 0, 0);
        if (NC > 1)
            for (int z = NSLOT + lane; z < NLDS; z += kWave) s_val[z] = 0.0;  // the further components' tiles
        __syncthreads();
    }

    // Software pipeline: the six stencil gathers of a step are issued at the END of the previous
    // step (right after relocation, together with the kappa gather), so they are in flight during
    // the whole deposit phase; slab-flush atomics produced by a step are issued in the NEXT step,
    // behind that step's gathers.
    double st_xp = 0, st_xm = 0, st_yp = 0, st_ym = 0, st_zp = 0, st_zm = 0;
    bool wave_on_face = true;  // wave-uniform: some lane's current cell lies on a grid face
    auto gather_stencil = [&]() {
        // :212-238 neighbours of the current node as table offsets.  Interior cells (every lane of
        // the wave, almost always) use the plain +-1 neighbours; the one-sided face rule is a rare,
        // wave-uniform branch.
        const bool on_face = (unsigned)(s.ci - 1) >= (unsigned)(nx - 2) || (unsigned)(s.cj - 1) >= (unsigned)(ny - 2) ||
                             (unsigned)(s.ck - 1) >= (unsigned)(nz - 2);
        wave_on_face = __builtin_amdgcn_ballot_w64(on_face) != 0ull;
        if (!wave_on_face) {
            // :254-265 six gathers from the node table, scalar strides added straight into the address
            st_xp = node_load<IDX64>(a.ne3d, cell + sX);
            st_xm = node_load<IDX64>(a.ne3d, cell - sX);
            st_yp = node_load<IDX64>(a.ne3d, cell + sY);
            st_ym = node_load<IDX64>(a.ne3d, cell - sY);
            st_zp = node_load<IDX64>(a.ne3d, cell + 1);
            st_zm = node_load<IDX64>(a.ne3d, cell - 1);
        } else {
            const int oxm = (s.ci == 0) ? 0 : ((s.ci == nx - 1) ? -2 * sX : -sX);
            const int oxp = (s.ci == 0) ? 2 * sX : ((s.ci == nx - 1) ? 0 : sX);
            const int oym = (s.cj == 0) ? 0 : ((s.cj == ny - 1) ? -2 * sY : -sY);
            const int oyp = (s.cj == 0) ? 2 * sY : ((s.cj == ny - 1) ? 0 : sY);
            const int ozm = (s.ck == 0) ? 0 : ((s.ck == nz - 1) ? -2 : -1);
            const int ozp = (s.ck == 0) ? 2 : ((s.ck == nz - 1) ? 0 : 1);
            st_xp = node_load<IDX64>(a.ne3d, cell + oxp);
            st_xm = node_load<IDX64>(a.ne3d, cell + oxm);
            st_yp = node_load<IDX64>(a.ne3d, cell + oyp);
            st_ym = node_load<IDX64>(a.ne3d, cell + oym);
            st_zp = node_load<IDX64>(a.ne3d, cell + ozp);
            st_zm = node_load<IDX64>(a.ne3d, cell + ozm);
        }
    };
    if (alive) gather_stencil();
    const double *const gk = CBET && a.gain ? a.gain + (long)beam * a.hsize : nullptr;  // this beam's gain grid
    double gained = 0.0;                     // CBET: energy this lane's ray gained
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;  // CBET = 4: the four field quantities this step deposits
    double dv0 = 0.0, dv1 = 0.0, dv2 = 0.0;  // deferred slab sums (one per axis) and their nodes
    int dn0 = 0, dn1 = 0, dn2 = 0;
    unsigned slabs_seen = 0;                 // wave-uniform: value of the slab counter when last drained

    for (int tt = 0; tt < a.nt; ++tt) {                        // :207
        const unsigned long long live_mask = __ballot(alive);
        if (live_mask == 0) break;
        w_steps_miss += 1u << 16;
        unsigned slot[8], node[8];
        double wgt[8];
        // Written and read by live lanes only (every later use is behind `alive`).  Dead lanes get "any
        // value" (a frozen unspecified value: well defined, and no per-step moves are spent on it).
        int hi, hj, hk, ax, ay, az;  // own node (haloed) and the lane's low corner
        int X0, X1, Y0, Y1, Z0, Z1;
        double fx, fy, fz, kap;
        hi = __builtin_nondeterministic_value(hi); hj = __builtin_nondeterministic_value(hj);
        hk = __builtin_nondeterministic_value(hk); ax = __builtin_nondeterministic_value(ax);
        ay = __builtin_nondeterministic_value(ay); az = __builtin_nondeterministic_value(az);
        X0 = __builtin_nondeterministic_value(X0); X1 = __builtin_nondeterministic_value(X1);
        Y0 = __builtin_nondeterministic_value(Y0); Y1 = __builtin_nondeterministic_value(Y1);
        Z0 = __builtin_nondeterministic_value(Z0); Z1 = __builtin_nondeterministic_value(Z1);
        fx = __builtin_nondeterministic_value(fx); fy = __builtin_nondeterministic_value(fy);
        fz = __builtin_nondeterministic_value(fz); kap = __builtin_nondeterministic_value(kap);
        if (alive) {
            // :268-273 kick then drift (stencil values gathered during the previous step)
            s.vx -= a.xconst * (st_xp - st_xm);
            s.vy -= a.yconst * (st_yp - st_ym);
            s.vz -= a.zconst * (st_zp - st_zm);
            s.px += s.vx * a.dt;
            s.py += s.vy * a.dt;
            s.pz += s.vz * a.dt;
            // :276-292 position in cell units, nearest-node update
            fx = (s.px - a.xmin) * a.inv_dx;
            fy = (s.py - a.ymin) * a.inv_dy;
            fz = (s.pz - a.zmin) * a.inv_dz;
            if (DEPOSIT == 3) {
                // fast path: interior cells, unambiguous matches (cbet_relocate.h); one ballot covers
                // all three axes, and `wave_on_face` was evaluated for this very cell by gather_stencil
                bool amb = false;
                const int qi = relocate_fast_interior(s.ci, fx, amb);
                const int qj = relocate_fast_interior(s.cj, fy, amb);
                const int qk = relocate_fast_interior(s.ck, fz, amb);
                if (wave_on_face || __builtin_amdgcn_ballot_w64(amb) != 0ull) {
                    s.ci = relocate_closed(s.ci, fx, nx);
                    s.cj = relocate_closed(s.cj, fy, ny);
                    s.ck = relocate_closed(s.ck, fz, nz);
                } else {
                    s.ci = qi;
                    s.cj = qj;
                    s.ck = qk;
                }
            } else {
                s.ci = relocate_loop(s.ci, fx, nx);
                s.cj = relocate_loop(s.cj, fy, ny);
                s.ck = relocate_loop(s.ck, fz, nz);
            }
            cell = (unsigned)(__mul24(__mul24(s.ci, ny) + s.cj, nz) + s.ck);
            // :296-298 absorption coefficient at the new node, then the NEXT step's stencil
            if (ABSORB) kap = node_load<IDX64>(a.kap3d, cell);
            gather_stencil();
        }
        if (DEPOSIT == 3 && (w_slabs_wide >> 16) != slabs_seen) {  // scalar test: a slab was retired last step
            // last step's retired slabs go to HBM now, younger than this step's gathers
            slabs_seen = w_slabs_wide >> 16;
#ifndef CBET_EXPERIMENT_DROP_FLUSH_ATOMICS
            if (dv0 != 0.0) { global_add(&edep[dn0], dv0); dv0 = 0.0; }
            if (dv1 != 0.0) { global_add(&edep[dn1], dv1); dv1 = 0.0; }
            if (dv2 != 0.0) { global_add(&edep[dn2], dv2); dv2 = 0.0; }
#else
            dv0 = dv1 = dv2 = 0.0;
#endif
        }
        if (alive) {
            // :299-311 absorbed energy.  For the windowed deposit the multiplication by `inc` is
            // applied last (below, after the window logic): kappa was requested a few instructions
            // ago and nothing from here to the deposit needs it, so a lone wave -- the tail of a
            // short launch -- no longer stalls on that gather every step.
            double inc = 0.0;
            if (DEPOSIT != 3) {
                if (ABSORB) {
                    inc = kap * s.uray;
                    s.uray -= inc;
                } else {
                    inc = s.uray;
                }
            }
            // :319-339 weights.  Each weight is (Fz * Fy) * Fx * inc with F = (1-d) for the ray's own
            // node along that axis and F = d for the neighbour on the `sign` side (:329-336).
            const double ox = fx - s.ci - 0.5, oy = fy - s.cj - 0.5, oz = fz - s.ck - 0.5;
            const double dm = 1.0 - fabs(ox), dn = 1.0 - fabs(oy), dl = 1.0 - fabs(oz);
            // :338-339 the neighbour lies on the side of the offset's sign, so a lane's two nodes per
            // axis are {low, low + 1} with low = own - 1 when the offset is negative
            const bool ngx = ox < 0, ngy = oy < 0, ngz = oz < 0;
            hi = s.ci + 1;
            hj = s.cj + 1;
            hk = s.ck + 1;
            ax = hi - (ngx ? 1 : 0);
            ay = hj - (ngy ? 1 : 0);
            az = hk - (ngz ? 1 : 0);
            // Corner order.  The eight (node, weight) pairs are the same whatever order they are
            // enumerated in, and every product keeps the reference's operand order.  With FLIP, three
            // lane bits swap which of an axis's two nodes is visited first, so rays a quarter cell apart
            // that share all 8 target nodes hit different nodes in any one ds_add_f64 instead of
            // serialising on one address.  Which bits: a patch row is lanes 8r..8r+7, and with 4 rays
            // per zone the 16 lanes of rows 0-3 x columns 0-3 share a cell.  Bits 0 and 1 (column) and
            // bit 3 (row) give those 16 lanes all 8 orders, two lanes each; bit 2 (the next cell over)
            // adds nothing (measured: z keyed on bit 2 21.9 ms, on bit 3 21.5 ms).
            const bool flx = FLIP && (lane & 1), fly = FLIP && (lane & 2), flz = FLIP && (lane & 8);
            const double ax_own = 1.0 - dm, ay_own = 1.0 - dn, az_own = 1.0 - dl;
            const double Fx0 = flx ? dm : ax_own, Fx1 = flx ? ax_own : dm;
            const double Fy0 = fly ? dn : ay_own, Fy1 = fly ? ay_own : dn;
            const double Fz0 = flz ? dl : az_own, Fz1 = flz ? az_own : dl;
            // first-visited node: the own node (high one iff the offset is negative) unless flipped
            const bool hx = ngx != flx, hy = ngy != fly, hz = ngz != flz;
            X0 = ax + (hx ? 1 : 0); X1 = ax + (hx ? 0 : 1);
            Y0 = ay + (hy ? 1 : 0); Y1 = ay + (hy ? 0 : 1);
            Z0 = az + (hz ? 1 : 0); Z1 = az + (hz ? 0 : 1);
            const double zy00 = Fz0 * Fy0, zy10 = Fz1 * Fy0, zy01 = Fz0 * Fy1, zy11 = Fz1 * Fy1;
            // order (x,y,z) = (0,0,0) (1,0,0) (0,0,1) (1,0,1) (0,1,0) (1,1,0) (0,1,1) (1,1,1) -- :341-348 without FLIP
            wgt[0] = zy00 * Fx0;
            wgt[1] = zy00 * Fx1;
            wgt[2] = zy10 * Fx0;
            wgt[3] = zy10 * Fx1;
            wgt[4] = zy01 * Fx0;
            wgt[5] = zy01 * Fx1;
            wgt[6] = zy11 * Fx0;
            wgt[7] = zy11 * Fx1;
            if (CBET) {
                // path length of the step; u_eff = the ray's energy averaged over the step
                double ds = 0.0;
                if (gk || CBET == 4) ds = sqrt(s.vx * s.vx + s.vy * s.vy + s.vz * s.vz) * a.dt;
                double u_eff = s.uray;
                if (gk) {
                    // K at the eight deposit nodes, weighted like the deposit.  The pairwise tree makes the
                    // sum independent of the corner order (FLIP swaps operands of commutative adds only).
                    const int nX0 = __mul24(X0, sXh), nX1 = __mul24(X1, sXh), nY0 = __mul24(Y0, sYh), nY1 = __mul24(Y1, sYh);
                    const double g0 = gain_load<IDX64>(gk, (unsigned)(nX0 + nY0 + Z0)), g1 = gain_load<IDX64>(gk, (unsigned)(nX1 + nY0 + Z0));
                    const double g2 = gain_load<IDX64>(gk, (unsigned)(nX0 + nY0 + Z1)), g3 = gain_load<IDX64>(gk, (unsigned)(nX1 + nY0 + Z1));
                    const double g4 = gain_load<IDX64>(gk, (unsigned)(nX0 + nY1 + Z0)), g5 = gain_load<IDX64>(gk, (unsigned)(nX1 + nY1 + Z0));
                    const double g6 = gain_load<IDX64>(gk, (unsigned)(nX0 + nY1 + Z1)), g7 = gain_load<IDX64>(gk, (unsigned)(nX1 + nY1 + Z1));
                    const double k01 = wgt[0] * g0 + wgt[1] * g1, k23 = wgt[2] * g2 + wgt[3] * g3;
                    const double k45 = wgt[4] * g4 + wgt[5] * g5, k67 = wgt[6] * g6 + wgt[7] * g7;
                    double x = ((k01 + k23) + (k45 + k67)) * ds;
                    if (x > a.max_exponent) x = a.max_exponent;
                    if (x < -a.max_exponent) x = -a.max_exponent;
                    const double phi = phi_det(x);
                    const double dg = s.uray * (x * phi);
                    u_eff = s.uray * phi;
                    gained += dg;
                    s.uray = s.uray + dg;
                }
                if (CBET == 4) {
                    q0 = u_eff * ds;
                    q1 = u_eff * (s.vx * a.dt);
                    q2 = u_eff * (s.vy * a.dt);
                    q3 = u_eff * (s.vz * a.dt);
                }
            }
            if (DEPOSIT != 3) {
#pragma unroll
                for (int c = 0; c < 8; ++c) wgt[c] = wgt[c] * inc;   // a_c * increment, :341-348
            }
            if (DEPOSIT != 3) {
                const int nX0 = __mul24(X0, sXh), nX1 = __mul24(X1, sXh), nY0 = __mul24(Y0, sYh), nY1 = __mul24(Y1, sYh);
                node[0] = nX0 + nY0 + Z0; node[1] = nX1 + nY0 + Z0; node[2] = nX0 + nY0 + Z1; node[3] = nX1 + nY0 + Z1;
                node[4] = nX0 + nY1 + Z0; node[5] = nX1 + nY1 + Z0; node[6] = nX0 + nY1 + Z1; node[7] = nX1 + nY1 + Z1;
            }
            if (DEPOSIT == 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) global_add(&edep[node[c]], wgt[c]);
                n_atomics += 8;
            }
            if (DEPOSIT == 2) {
                slot[0] = tagged.slot(X0, Y0, Z0);
                slot[1] = tagged.slot(X1, Y0, Z0);
                slot[2] = tagged.slot(X0, Y0, Z1);
                slot[3] = tagged.slot(X1, Y0, Z1);
                slot[4] = tagged.slot(X0, Y1, Z0);
                slot[5] = tagged.slot(X1, Y1, Z0);
                slot[6] = tagged.slot(X0, Y1, Z1);
                slot[7] = tagged.slot(X1, Y1, Z1);
            }
            ++nsteps;
        }
        if (DEPOSIT == 2) lds_deposit8<WL>(tagged, alive, slot, node, wgt, edep, n_evict);
        if (DEPOSIT == 3) {
            // the lane's 8 targets span {low, low + 1} per axis; (ax, ay, az) is its low corner
            int wide = 0;
            using MW = MovingWindow<WL, RL>;
            bool inbox;            // the lane deposits into LDS this step
            int tile = 0;          // ... into this tile (offset in doubles)
            if (!TWOBOX) {
                win.template follow_axis<0, true>(win.ox, alive, ax, lane, edep, sXh, sYh, n_atomics, wide, w_slabs_wide, dv0, dn0);
                win.template follow_axis<1, true>(win.oy, alive, ay, lane, edep, sXh, sYh, n_atomics, wide, w_slabs_wide, dv1, dn1);
                win.template follow_axis<2, true>(win.oz, alive, az, lane, edep, sXh, sYh, n_atomics, wide, w_slabs_wide, dv2, dn2);
                __builtin_amdgcn_wave_barrier();
                inbox = alive && win.holds(ax, ay, az);
            } else {
                // box A follows the lanes whose home it is
                const bool memA = alive && !homeB;
                win.template follow_axis<0, true>(win.ox, memA, ax, lane, edep, sXh, sYh, n_atomics, wide, w_slabs_wide, dv0, dn0);
                win.template follow_axis<1, true>(win.oy, memA, ay, lane, edep, sXh, sYh, n_atomics, wide, w_slabs_wide, dv1, dn1);
                win.template follow_axis<2, true>(win.oz, memA, az, lane, edep, sXh, sYh, n_atomics, wide, w_slabs_wide, dv2, dn2);
                const bool inA = alive && win.holds(ax, ay, az);
                bool inB = false;
                if (b_active) {  // scalar branch
                    w_slabs_wide += 1u;
                    const bool memB = alive && homeB;
                    double tv = 0.0;
                    int tn = 0, tw = 0;
                    winB.template follow_axis<0, false>(winB.ox, memB, ax, lane, edep, sXh, sYh, n_atomics, tw, w_slabs_wide, tv, tn);
                    winB.template follow_axis<1, false>(winB.oy, memB, ay, lane, edep, sXh, sYh, n_atomics, tw, w_slabs_wide, tv, tn);
                    winB.template follow_axis<2, false>(winB.oz, memB, az, lane, edep, sXh, sYh, n_atomics, tw, w_slabs_wide, tv, tn);
                    inB = alive && winB.holds(ax, ay, az);
                }
                // lanes that fell out of A look for a home in B; an idle B is re-created around the first of them
                const bool lost = alive && !homeB && !inA;
                const unsigned long long lost_mask = __builtin_amdgcn_ballot_w64(lost);
                if (lost_mask != 0ull) {
                    if (!b_active) {
                        const int src = __ffsll((long long)lost_mask) - 1;
                        winB.ox = __builtin_amdgcn_readlane(ax, src) - (W / 2 - 1);
                        winB.oy = __builtin_amdgcn_readlane(ay, src) - (W / 2 - 1);
                        winB.oz = __builtin_amdgcn_readlane(az, src) - (W / 2 - 1);
                        b_active = true;  // its tile is all zero: zeroed at start and flushed whenever it empties
                        inB = alive && winB.holds(ax, ay, az);
                    }
                    homeB = homeB || (lost && inB);
                }
                if (b_active) {
                    // a B lane that drifted out of B but back into A goes home
                    if (alive && homeB && !inB && inA) homeB = false;
                    if (__builtin_amdgcn_ballot_w64(alive && homeB) == 0ull) {
                        __builtin_amdgcn_wave_barrier();
                        winB.flush_all(lane, edep, sXh, sYh, n_atomics);
                        b_active = false;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                const bool useB = alive && homeB && inB;
                inbox = useB || (alive && !homeB && inA);
                tile = useB ? MW::NDOUBLES : 0;
            }
            if (__builtin_amdgcn_ballot_w64(alive && !inbox) != 0ull) {
                w_steps_miss += 1u;
                if (!TWOBOX && wide) w_slabs_wide += 1u;
            }
            if (alive) {  // :305-311, then a_c * increment (:341-348)
                double inc;
                if (ABSORB) {
                    inc = kap * s.uray;
                    s.uray -= inc;
                } else {
                    inc = s.uray;
                }
                if (CBET == 4) inc = q0;
#pragma unroll
                for (int c = 0; c < 8; ++c) wgt[c] = wgt[c] * inc;
            }
            // key: identifies the ordered set of 8 target nodes (own node + the three signs); lanes may
            // only be merged when they enumerate the corners in the same order, i.e. without FLIP
            int key = inbox ? 0 : -2 - lane;
            if (PRE >= 1 && !FLIP) {
                if (inbox)
                    key = (int)(((__mul24(hi, sXh) + __mul24(hj, sYh) + hk) << 3) | ((hi - ax) << 2) | ((hj - ay) << 1) | (hk - az));
                merge_level<0xB1, 1>(lane, key, wgt);                 // quad_perm [1,0,3,2]: lane ^ 1
                if (PRE >= 2) merge_level<0x4E, 2>(lane, key, wgt);   // quad_perm [2,3,0,1]: lane ^ 2
            }
            if (key >= 0) {
                const int copy = ((lane >> PRE) & (MW::R - 1)) * MW::CS + tile;
                const int x0 = (X0 & (W - 1)) * MW::XS + copy, x1 = (X1 & (W - 1)) * MW::XS + copy;
                const int y0 = (Y0 & (W - 1)) * MW::YS, y1 = (Y1 & (W - 1)) * MW::YS;
                const int z0 = Z0 & (W - 1), z1 = Z1 & (W - 1);
                win.add(x0 + y0 + z0, wgt[0]);
                win.add(x1 + y0 + z0, wgt[1]);
                win.add(x0 + y0 + z1, wgt[2]);
                win.add(x1 + y0 + z1, wgt[3]);
                win.add(x0 + y1 + z0, wgt[4]);
                win.add(x1 + y1 + z0, wgt[5]);
                win.add(x0 + y1 + z1, wgt[6]);
                win.add(x1 + y1 + z1, wgt[7]);
            } else if (alive && !inbox) {
#ifndef CBET_EXPERIMENT_DROP_MISS_ATOMICS  // timing-only experiment builds; never shipped
                const int nX0 = __mul24(X0, sXh), nX1 = __mul24(X1, sXh), nY0 = __mul24(Y0, sYh), nY1 = __mul24(Y1, sYh);
                global_add(&edep[nX0 + nY0 + Z0], wgt[0]);
                global_add(&edep[nX1 + nY0 + Z0], wgt[1]);
                global_add(&edep[nX0 + nY0 + Z1], wgt[2]);
                global_add(&edep[nX1 + nY0 + Z1], wgt[3]);
                global_add(&edep[nX0 + nY1 + Z0], wgt[4]);
                global_add(&edep[nX1 + nY1 + Z0], wgt[5]);
                global_add(&edep[nX0 + nY1 + Z1], wgt[6]);
                global_add(&edep[nX1 + nY1 + Z1], wgt[7]);
#endif
                n_atomics += 8;
                ++n_evict;  // counted as "ray-steps that missed the window"
            }
            if (CBET == 4 && alive) {
                // Displacement components: the ray's own node only -- box A's tiles, or HBM for a lane of
                // box B / outside the boxes.  (Merging the four rays of a quad in registers first, which
                // quarters the same-address LDS adds, changed nothing: 49.9 ms either way.)
                if (inbox && tile == 0) {
                    const int own = MW::addr_d(hi & (W - 1), hj & (W - 1), hk & (W - 1)) + NSLOT;
                    win.add(own, q1);
                    win.add(own + MW::DT, q2);
                    win.add(own + 2 * MW::DT, q3);
                } else {
                    const int own = __mul24(hi, sXh) + __mul24(hj, sYh) + hk;
#ifndef CBET_EXPERIMENT_DROP_MISS_ATOMICS
                    global_add(&edep[a.comp_stride + own], q1);
                    global_add(&edep[2 * a.comp_stride + own], q2);
                    global_add(&edep[3 * a.comp_stride + own], q3);
#endif
                    n_atomics += 3;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // :351-356.  The six box bounds are compared only when some lane is within two cells of a
        // face (wave-uniform ballot on the cell-unit position): a ray with 2 < f < n-3 on every axis
        // is more than a cell and a half inside [min - d/2, max + d/2], far beyond any rounding.
        // This keeps 12 scalar registers out of the hot loop (they are re-read from the argument
        // block in the rare branch).
        if (alive && s.uray <= s.ustop) alive = false;
        const bool near_face = alive && !(fx > 2.0 && fx < a.fx_hi && fy > 2.0 && fy < a.fy_hi && fz > 2.0 && fz < a.fz_hi);
        if (__builtin_amdgcn_ballot_w64(near_face) != 0ull) {
            const double *b = a.bounds;  // {xlo, xhi, ylo, yhi, zlo, zhi}
            if (alive && (s.px < b[0] || s.px > b[1] || s.py < b[2] || s.py > b[3] || s.pz < b[4] || s.pz > b[5]))
                alive = false;
        }
    }

    if (DEPOSIT == 2) {
        __syncthreads();
        n_atomics += tagged.flush(lane, edep) + n_evict;
    }
    if (DEPOSIT == 3) {
        if (dv0 != 0.0) global_add(&edep[dn0], dv0);
        if (dv1 != 0.0) global_add(&edep[dn1], dv1);
        if (dv2 != 0.0) global_add(&edep[dn2], dv2);
        __syncthreads();
        win.flush_all(lane, edep, sXh, sYh, n_atomics);
        if (TWOBOX && b_active) winB.flush_all(lane, edep, sXh, sYh, n_atomics);
    }
#ifdef CBET_EXPERIMENT_TIMELINE
    if (lane == 0 && a.timeline) {
        a.timeline[3 * (long)blockIdx.x + 0] = t_start;
        a.timeline[3 * (long)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        a.timeline[3 * (long)blockIdx.x + 2] = (unsigned long long)(w_steps_miss >> 16);
    }
#endif
    if (CBET && a.beam_gain) {  // one fp64 atomic per wave
        double t = gained;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, kWave);
        if (lane == 0 && t != 0.0) atomicAdd(&a.beam_gain[beam], t);
    }
    // counters: one atomic per wave and counter
    const int tot_steps = wave_sum(nsteps), tot_rays = wave_sum(launched), tot_at = wave_sum(n_atomics),
              tot_ev = wave_sum(n_evict);
    if (lane == 0) {
        atomicAdd(&a.counters[kCntSteps], (unsigned long long)tot_steps);
        atomicAdd(&a.counters[kCntRays], (unsigned long long)tot_rays);
        atomicAdd(&a.counters[kCntGlobalAtomics], (unsigned long long)tot_at);
        atomicAdd(&a.counters[kCntEvictions], (unsigned long long)tot_ev);
        atomicAdd(&a.counters[kCntWaveSteps], (unsigned long long)(w_steps_miss >> 16));
        if (DEPOSIT == 3) {
            atomicAdd(&a.counters[kCntWaveStepsMiss], (unsigned long long)(w_steps_miss & 0xFFFFu));
            atomicAdd(&a.counters[kCntWaveStepsWide], (unsigned long long)(w_slabs_wide & 0xFFFFu));
            atomicAdd(&a.counters[kCntSlabsRetired], (unsigned long long)(w_slabs_wide >> 16));
        }
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

template <int DEPOSIT, int WL, int RL, int PRE, bool FLIP, bool TWOBOX, bool IDX64>
static void launch_k(const TraceArgs &a, dim3 grid, hipStream_t stream)
{
    if constexpr (DEPOSIT == 3 && FLIP && RL == 0 && PRE == 0) {  // CBET hooks: the two-box default and its one-box sibling
      if (a.quantity != 0) {  // the fused four-component field pass
        if (a.absorption == 1)
            hipLaunchKernelGGL((k_trace<DEPOSIT, WL, RL, PRE, FLIP, TWOBOX, IDX64, true, 4>), grid, dim3(kWave), 0, stream, a);
        else
            hipLaunchKernelGGL((k_trace<DEPOSIT, WL, RL, PRE, FLIP, TWOBOX, IDX64, false, 4>), grid, dim3(kWave), 0, stream, a);
        return;
      }
      if (a.gain || a.beam_gain) {
        if (a.absorption == 1)
            hipLaunchKernelGGL((k_trace<DEPOSIT, WL, RL, PRE, FLIP, TWOBOX, IDX64, true, 1>), grid, dim3(kWave), 0, stream, a);
        else
            hipLaunchKernelGGL((k_trace<DEPOSIT, WL, RL, PRE, FLIP, TWOBOX, IDX64, false, 1>), grid, dim3(kWave), 0, stream, a);
        return;
      }
    }
    if (a.absorption == 1)
        hipLaunchKernelGGL((k_trace<DEPOSIT, WL, RL, PRE, FLIP, TWOBOX, IDX64, true>), grid, dim3(kWave), 0, stream, a);
    else
        hipLaunchKernelGGL((k_trace<DEPOSIT, WL, RL, PRE, FLIP, TWOBOX, IDX64, false>), grid, dim3(kWave), 0, stream, a);
}

template <bool IDX64>
static void dispatch_trace(const TraceArgs &a, int variant, int wl, int rl, int pre, bool flip, bool twobox,
                           dim3 grid, hipStream_t stream)
{
    if (variant == CBET_KERNEL_GLOBAL_ATOMICS) {
        launch_k<1, 1, 0, 0, false, false, IDX64>(a, grid, stream);
    } else if (variant == CBET_KERNEL_LDS_COMBINE) {
        if (wl == 4) launch_k<2, 4, 0, 0, false, false, IDX64>(a, grid, stream);
        else launch_k<2, 3, 0, 0, false, false, IDX64>(a, grid, stream);
    } else if (wl == 4) {
        launch_k<3, 4, 0, 0, false, false, IDX64>(a, grid, stream);
    } else if (twobox) {
        launch_k<3, 3, 0, 0, true, true, IDX64>(a, grid, stream);
    } else if (flip) {
        switch (rl) {
        case 0: launch_k<3, 3, 0, 0, true, false, IDX64>(a, grid, stream); break;
        case 1: launch_k<3, 3, 1, 0, true, false, IDX64>(a, grid, stream); break;
        default: launch_k<3, 3, 2, 0, true, false, IDX64>(a, grid, stream); break;
        }
    } else {
        switch (rl * 3 + pre) {
        case 0: launch_k<3, 3, 0, 0, false, false, IDX64>(a, grid, stream); break;
        case 1: launch_k<3, 3, 0, 1, false, false, IDX64>(a, grid, stream); break;
        case 2: launch_k<3, 3, 0, 2, false, false, IDX64>(a, grid, stream); break;
        case 3: launch_k<3, 3, 1, 0, false, false, IDX64>(a, grid, stream); break;
        case 4: launch_k<3, 3, 1, 1, false, false, IDX64>(a, grid, stream); break;
        case 5: launch_k<3, 3, 1, 2, false, false, IDX64>(a, grid, stream); break;
        case 6: launch_k<3, 3, 2, 0, false, false, IDX64>(a, grid, stream); break;
        case 7: launch_k<3, 3, 2, 1, false, false, IDX64>(a, grid, stream); break;
        default: launch_k<3, 3, 2, 2, false, false, IDX64>(a, grid, stream); break;
        }
    }
}

hipError_t audit_violations(unsigned long long *out, bool reset, hipStream_t stream)
{
#ifdef CBET_DEBUG_BOUNDS
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_audit_violations), sizeof *out);
    if (e == hipSuccess && reset) {
        const unsigned long long zero = 0;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_audit_violations), &zero, sizeof zero);
    }
    return e;
#else
    (void)out; (void)reset; (void)stream;
    return hipErrorNotSupported;
#endif
}

hipError_t launch_trace(const TraceArgs &a, int variant, int window_log2, int copies_log2, int prereduce,
                        bool corner_flip, bool two_boxes, bool force_idx64, hipStream_t stream)
{
    const long waves = (a.total_bundles - a.shard_index + a.shard_count - 1) / a.shard_count;
    if (waves <= 0) return hipSuccess;
#ifdef CBET_DEBUG_BOUNDS
    {
        const long cells = (long)(a.nx + 2) * (a.ny + 2) * (a.nz + 2);
        const double *lo = a.edep, *hi = a.edep + (a.grid_stride ? a.grid_stride * (long)(a.beam_lo + a.nbeams_local) : cells);
        if (a.quantity != 0) hi = a.edep + 4 * a.comp_stride;   // the field pass writes four component arrays
        const unsigned long long nodes = (unsigned long long)a.nx * a.ny * a.nz, hs = (unsigned long long)a.hsize;
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_audit_hsize), &hs, sizeof hs, 0, hipMemcpyHostToDevice, stream);
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_audit_edep_lo), &lo, sizeof lo, 0, hipMemcpyHostToDevice, stream);
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_audit_edep_hi), &hi, sizeof hi, 0, hipMemcpyHostToDevice, stream);
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_audit_nodes), &nodes, sizeof nodes, 0, hipMemcpyHostToDevice, stream);
    }
#endif
    const long chunk = a.xcd_chunk > 0 ? (waves + 7) / 8 : 0;
    TraceArgs b = a;
    b.xcd_chunk = (int)chunk;
    const dim3 grid((unsigned)(chunk > 0 ? chunk * 8 : waves));
    // 32-bit byte offsets into the node tables -- and, with the CBET hooks, into a beam's haloed gain grid
    const unsigned long long table_bytes = 8ull * (a.gain ? (unsigned long long)a.hsize : (unsigned long long)a.nx * a.ny * a.nz);
    const bool idx64 = force_idx64 || table_bytes >= (1ull << 32);
    // the pre-reduction key packs (flat haloed node index << 3 | signs) into 31 bits
    if ((long)(a.nx + 2) * (a.ny + 2) * (a.nz + 2) >= (1L << 28)) prereduce = 0;
    const bool flip = prereduce == 0 && corner_flip;
    if (idx64) dispatch_trace<true>(b, variant, window_log2, copies_log2, prereduce, flip, two_boxes, grid, stream);
    else dispatch_trace<false>(b, variant, window_log2, copies_log2, prereduce, flip, two_boxes, grid, stream);
    return hipGetLastError();
}

}  // namespace cbet
