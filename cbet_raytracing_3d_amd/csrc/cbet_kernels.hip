// cbet_kernels.hip -- gfx950 (CDNA4) kernels of the ray-integrator path: the node-table kernel and the two
// cross-check formulations of the ray integrator.  The shipped integrator is cbet_trace_window.hip.
//
// Built with -ffp-contract=off: every per-ray fp64 operation below is the single IEEE operation the
// reference's statement performs (/root/reference/launch_ray_XZ.cu:117-359), in the same order, so
// a ray's trajectory, absorbed energy and the eight deposit values are the values the CPU oracle
// computes; only the order in which different rays' deposits are summed is free.
//
//   * k_tabulate     : the radial (r, ne, Te) profile is evaluated ONCE per node into two node
//                      tables in HBM, ne3d and kappa3d (= ed/ncrit*nuei*dt, launch_ray_XZ.cu:296-305
//                      without the trailing *uray).  The reference re-interpolates the profile eight
//                      times per ray-step (8 bisections + 9 sqrt + 9 div); with the tables a ray-step is
//                      seven 8-byte gathers and ~60 flops, no sqrt/div (k_trace_simple) -- or, folded once
//                      more into one 32-byte record per node by k_step_table, a single gather (the shipped kernel).
//   * k_trace_simple : one wavefront = one 8x8-ray patch, the reference's step loop written plainly
//                      (literal relocation loop, no software pipeline), with the deposit either as
//       1  GLOBAL : 8 global_atomic_add_f64 per ray-step -- the reference's own scheme, the baseline; or
//       2  TAGGED : a wave-private toroidal LDS tile with node tags; slots are claimed by LDS CAS and
//                   written back with one global atomic when another node claims them.
//     Both exist to cross-check the shipped kernel (tests/) and to price its deposit scheme.
#include <hip/hip_runtime.h>

#include "cbet_trace_common.h"

namespace cbet {
namespace {

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
// Step records (cbet_device.h StepRecord): one thread per node, z fastest.  Bound: HBM, 16 B read (plus
// neighbour lines from cache) and 32 B written per node.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_step_table(const StepTableArgs a)
{
    const long total = (long)a.nx * a.ny * a.nz;
    const long stride = (long)gridDim.x * blockDim.x;
    const long sY = a.nz, sX = (long)a.ny * a.nz;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int k = (int)(idx % a.nz);
        const long ij = idx / a.nz;
        const int j = (int)(ij % a.ny);
        const int i = (int)(ij / a.ny);
        // launch_ray_XZ.cu:212-238 : neighbours this+-1, one-sided on the faces (0 -> (0, 2), n-1 -> (n-3, n-1))
        const long oxm = (i == 0) ? 0 : ((i == a.nx - 1) ? -2 * sX : -sX), oxp = (i == 0) ? 2 * sX : ((i == a.nx - 1) ? 0 : sX);
        const long oym = (j == 0) ? 0 : ((j == a.ny - 1) ? -2 * sY : -sY), oyp = (j == 0) ? 2 * sY : ((j == a.ny - 1) ? 0 : sY);
        const long ozm = (k == 0) ? 0 : ((k == a.nz - 1) ? -2 : -1), ozp = (k == 0) ? 2 : ((k == a.nz - 1) ? 0 : 1);
        StepRecord r;
        r.kx = a.xconst * (a.ne3d[idx + oxp] - a.ne3d[idx + oxm]);   // :268
        r.ky = a.yconst * (a.ne3d[idx + oyp] - a.ne3d[idx + oym]);   // :269
        r.kz = a.zconst * (a.ne3d[idx + ozp] - a.ne3d[idx + ozm]);   // :270
        r.kap = a.kap3d[idx];
        a.rec[idx] = r;
    }
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
    const TraceArgs *args;

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
        if (CBET_AUDIT(*args, s < (unsigned)NSLOT))
            __hip_atomic_fetch_add(&val[s], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // Write every live slot back (wave end).
    __device__ __forceinline__ int flush(int lane, double *edep)
    {
        int n = 0;
        for (int s = lane; s < NSLOT; s += kWave) {
            const unsigned t = tag[s];
            if (t != kEmptyTag) {
                global_add(*args, &edep[t], val[s]);
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
                        global_add(*win.args, &edep[t], __longlong_as_double((long long)old));
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
// The cross-check integrator.  DEPOSIT 1: 8 global atomics per step; 2: tagged 8^3 LDS window.
// Absorption (def.cuh:118) and the table index width are run-time here: nothing in it is tuned.
// ---------------------------------------------------------------------------------------------
template <int DEPOSIT>
__global__ void __launch_bounds__(kWave) k_trace_simple(const TraceArgs a)
{
    constexpr int WL = 3;
    constexpr int NSLOT = DEPOSIT == 2 ? (1 << (3 * WL)) : 1;
    __shared__ double s_val[NSLOT];
    __shared__ unsigned s_tag[NSLOT];
    const int lane = threadIdx.x;
    int beam, patch;
    if (!work_item(a, blockIdx.x, beam, patch)) return;
    double *const edep = a.edep + (long)(beam - a.grid_beam0) * a.grid_stride;
    const bool absorb = a.absorption == 1;

    Ray s;
    const int li = patch * kWave + lane;
    const int pre_raynum = li < a.nlive ? a.live[li] : -1;  // -1: hole in the 8x8 patch
    bool alive = pre_raynum >= 0;
    if (alive) alive = launch_ray(a, beam, pre_raynum, s);
    const int launched = alive ? 1 : 0;

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const long sY = nz, sX = (long)ny * nz;                       // node-table strides (elements)
    const int sYh = a.sYh, sXh = a.sXh;                           // haloed edep strides (:5-7)
    int nsteps = 0, n_atomics = 0, n_evict = 0;
    unsigned wave_steps = 0;

    LdsWindow<WL> tagged{s_val, s_tag, &a};
    if (DEPOSIT == 2) {
        tagged.clear(lane);
        __syncthreads();
    }

    for (int tt = 0; tt < a.nt; ++tt) {                        // :207
        if (__ballot(alive) == 0) break;
        ++wave_steps;
        unsigned slot[8] = {0, 0, 0, 0, 0, 0, 0, 0}, node[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double wgt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (alive) {
            // :212-238 neighbours of the current node, one-sided on the faces
            const int im = (s.ci == 0) ? 0 : ((s.ci == nx - 1) ? nx - 3 : s.ci - 1);
            const int ip = (s.ci == 0) ? 2 : ((s.ci == nx - 1) ? nx - 1 : s.ci + 1);
            const int jm = (s.cj == 0) ? 0 : ((s.cj == ny - 1) ? ny - 3 : s.cj - 1);
            const int jp = (s.cj == 0) ? 2 : ((s.cj == ny - 1) ? ny - 1 : s.cj + 1);
            const int km = (s.ck == 0) ? 0 : ((s.ck == nz - 1) ? nz - 3 : s.ck - 1);
            const int kp = (s.ck == 0) ? 2 : ((s.ck == nz - 1) ? nz - 1 : s.ck + 1);
            auto ne_at = [&](int i, int j, int k) { return node_load<true>(a, a.ne3d, (unsigned)(i * sX + j * sY + k)); };
            // :254-273 six gathers, kick, drift
            s.vx -= a.xconst * (ne_at(ip, s.cj, s.ck) - ne_at(im, s.cj, s.ck));
            s.vy -= a.yconst * (ne_at(s.ci, jp, s.ck) - ne_at(s.ci, jm, s.ck));
            s.vz -= a.zconst * (ne_at(s.ci, s.cj, kp) - ne_at(s.ci, s.cj, km));
            s.px += s.vx * a.dt;
            s.py += s.vy * a.dt;
            s.pz += s.vz * a.dt;
            // :276-292 position in cell units, nearest-node update (the literal loop)
            const double fx = (s.px - a.xmin) * a.inv_dx;
            const double fy = (s.py - a.ymin) * a.inv_dy;
            const double fz = (s.pz - a.zmin) * a.inv_dz;
            s.ci = relocate_loop(s.ci, fx, nx);
            s.cj = relocate_loop(s.cj, fy, ny);
            s.ck = relocate_loop(s.ck, fz, nz);
            // :296-311 absorbed energy
            double inc;
            if (absorb) {
                inc = node_load<true>(a, a.kap3d, (unsigned)(s.ci * sX + s.cj * sY + s.ck)) * s.uray;
                s.uray -= inc;
            } else {
                inc = s.uray;
            }
            // :319-339 weights, in the reference's corner order
            const double ox = fx - s.ci - 0.5, oy = fy - s.cj - 0.5, oz = fz - s.ck - 0.5;
            const double dm = 1.0 - fabs(ox), dn = 1.0 - fabs(oy), dl = 1.0 - fabs(oz);
            const int X0 = s.ci + 1, X1 = X0 + (ox < 0 ? -1 : 1);
            const int Y0 = s.cj + 1, Y1 = Y0 + (oy < 0 ? -1 : 1);
            const int Z0 = s.ck + 1, Z1 = Z0 + (oz < 0 ? -1 : 1);
            const double Fx0 = 1.0 - dm, Fy0 = 1.0 - dn, Fz0 = 1.0 - dl;
            const double zy00 = Fz0 * Fy0, zy10 = dl * Fy0, zy01 = Fz0 * dn, zy11 = dl * dn;
            wgt[0] = zy00 * Fx0 * inc; wgt[1] = zy00 * dm * inc; wgt[2] = zy10 * Fx0 * inc; wgt[3] = zy10 * dm * inc;
            wgt[4] = zy01 * Fx0 * inc; wgt[5] = zy01 * dm * inc; wgt[6] = zy11 * Fx0 * inc; wgt[7] = zy11 * dm * inc;
            const int nX0 = X0 * sXh, nX1 = X1 * sXh, nY0 = Y0 * sYh, nY1 = Y1 * sYh;
            node[0] = nX0 + nY0 + Z0; node[1] = nX1 + nY0 + Z0; node[2] = nX0 + nY0 + Z1; node[3] = nX1 + nY0 + Z1;
            node[4] = nX0 + nY1 + Z0; node[5] = nX1 + nY1 + Z0; node[6] = nX0 + nY1 + Z1; node[7] = nX1 + nY1 + Z1;
            if (DEPOSIT == 1) {
#pragma unroll
                for (int c = 0; c < 8; ++c) global_add(a, &edep[node[c]], wgt[c]);   // :341-348
                n_atomics += 8;
            } else {
                slot[0] = tagged.slot(X0, Y0, Z0); slot[1] = tagged.slot(X1, Y0, Z0);
                slot[2] = tagged.slot(X0, Y0, Z1); slot[3] = tagged.slot(X1, Y0, Z1);
                slot[4] = tagged.slot(X0, Y1, Z0); slot[5] = tagged.slot(X1, Y1, Z0);
                slot[6] = tagged.slot(X0, Y1, Z1); slot[7] = tagged.slot(X1, Y1, Z1);
            }
            ++nsteps;
        }
        if (DEPOSIT == 2) lds_deposit8<WL>(tagged, alive, slot, node, wgt, edep, n_evict);
        if (alive) {                                               // :351-356
            const double *b = a.bounds;  // {xlo, xhi, ylo, yhi, zlo, zhi}
            if (s.uray <= s.ustop || s.px < b[0] || s.px > b[1] || s.py < b[2] || s.py > b[3] || s.pz < b[4] || s.pz > b[5])
                alive = false;
        }
    }
    if (DEPOSIT == 2) {
        __syncthreads();
        n_atomics += tagged.flush(lane, edep) + n_evict;
    }
    const int tot_steps = wave_sum(nsteps), tot_rays = wave_sum(launched), tot_at = wave_sum(n_atomics),
              tot_ev = wave_sum(n_evict);
    if (lane == 0) {
        atomicAdd(&a.counters[kCntSteps], (unsigned long long)tot_steps);
        atomicAdd(&a.counters[kCntRays], (unsigned long long)tot_rays);
        atomicAdd(&a.counters[kCntGlobalAtomics], (unsigned long long)tot_at);
        atomicAdd(&a.counters[kCntEvictions], (unsigned long long)tot_ev);
        atomicAdd(&a.counters[kCntWaveSteps], (unsigned long long)wave_steps);
    }
}

#ifdef CBET_DEBUG_BOUNDS
__device__ unsigned long long g_audit_violations;
#endif

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

hipError_t launch_step_table(const StepTableArgs &a, hipStream_t stream)
{
    const long total = (long)a.nx * a.ny * a.nz;
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_step_table, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
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

hipError_t launch_trace(const TraceArgs &a0, int variant, bool force_idx64, hipStream_t stream)
{
    TraceArgs a = a0;
#ifdef CBET_DEBUG_BOUNDS
    {   // the ranges the audited accesses are checked against
        const long cells = (long)(a.nx + 2) * a.sXh;
        a.audit_lo = a.edep;
        a.audit_hi = a.edep + (a.grid_stride ? a.grid_stride * (long)(a.beam_lo - a.grid_beam0 + a.nbeams_local) : cells);
        if (a.quantity != 0) a.audit_hi = a.edep + 4 * a.comp_stride;   // the field pass writes four component arrays
        a.audit_nodes = (unsigned long long)a.nx * a.ny * a.nz;
        a.audit_hsize = (unsigned long long)a.hsize;
        hipError_t e = hipGetSymbolAddress((void **)&a.audit_count, HIP_SYMBOL(g_audit_violations));
        if (e != hipSuccess) return e;
    }
#endif
    if (variant == CBET_KERNEL_LDS_WINDOW) return launch_trace_window(a, force_idx64, stream);
    const long waves = a.item_count;
    if (waves <= 0) return hipSuccess;
    const dim3 grid((unsigned)waves);
    if (variant == CBET_KERNEL_GLOBAL_ATOMICS) hipLaunchKernelGGL((k_trace_simple<1>), grid, dim3(kWave), 0, stream, a);
    else hipLaunchKernelGGL((k_trace_simple<2>), grid, dim3(kWave), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cbet
