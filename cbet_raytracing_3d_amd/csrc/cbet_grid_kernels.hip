// cbet_grid_kernels.hip -- the cell-parallel kernels beside the ray integrator (cbet_kernels.hip):
//   * k_gain_field, k_gain_field_sym : per-beam fields -> CBET gain coefficient (no
//     reference counterpart, parity unpinned -- model and layout in DESIGN.md section 9)
//   * k_edep_average                 : the 27-point node average of main.cu:334-349
// Built with -ffp-contract=off like the rest of the library: every statement below is one IEEE
// operation in the order written, which is the order the CPU checker of the CBET stage uses.
#include <hip/hip_runtime.h>

#include "cbet_device.h"

namespace cbet {
namespace {

// ---------------------------------------------------------------------------------------------
// CBET extension (no reference counterpart; model and layout in DESIGN.md section 9).
// Deposit-grid cell (hi,hj,hk) takes its plasma state from node (hi-1,hj-1,hk-1), clamped.
// ---------------------------------------------------------------------------------------------
struct CellState {
    double frac, eps, rt, ux, uy, uz;   // ne/ncrit, 1 - ne/ncrit, sqrt(eps), flow velocity
};

__device__ __forceinline__ CellState cell_state(const GainArgs &a, long h)
{
    const int sYh = a.nz + 2;
    const long sXh = (long)(a.ny + 2) * sYh;
    const int hi = (int)(h / sXh);
    const int rem = (int)(h - hi * sXh);
    const int hj = rem / sYh, hk = rem - hj * sYh;
    const int i = hi < 1 ? 0 : (hi > a.nx ? a.nx - 1 : hi - 1);
    const int j = hj < 1 ? 0 : (hj > a.ny ? a.ny - 1 : hj - 1);
    const int k = hk < 1 ? 0 : (hk > a.nz ? a.nz - 1 : hk - 1);
    CellState c;
    c.frac = a.ne3d[((long)i * a.ny + j) * a.nz + k] / a.ncrit;
    c.eps = 1.0 - c.frac;
    c.rt = c.eps > 0.0 ? sqrt(c.eps) : 0.0;
    const double xc = i * a.dx + a.xmin, yc = j * a.dy + a.ymin, zc = k * a.dz + a.zmin;
    const double rr = sqrt(xc * xc + yc * yc + zc * zc);
    double t = (rr - a.mach_r0) / (a.mach_r1 - a.mach_r0);
    if (t < 0.0) t = 0.0;
    if (t > 1.0) t = 1.0;
    const double um = (a.mach_0 + (a.mach_1 - a.mach_0) * t) * a.cs;
    c.ux = c.uy = c.uz = 0.0;
    if (rr > 0.0) { c.ux = um * (xc / rr); c.uy = um * (yc / rr); c.uz = um * (zc / rr); }
    return c;
}

// Phase 1 of both gain kernels for one beam's entry of a cell that the beam's rays touched (E != 0): the deposited
// (E, Dx, Dy, Dz) become (I, kx, ky, kz) in place.  I = E / (group speed x dt) where the beam is present (E > 0,
// sub-critical plasma, a direction known), else 0; k = |k| D / |D|.  With GainArgs.frozen the three direction
// entries already hold k -- written by an earlier call on the fields of the gain-free first pass -- and only the
// energy entry is read and replaced.
__device__ __forceinline__ void normalise_entry(const GainArgs &a, const CellState &c, double kmag, double ds_node,
                                                double E, double *fI, double *fx, double *fy, double *fz, long o)
{
    double I = 0.0;
    if (a.frozen) {
        const double kx = fx[o], ky = fy[o], kz = fz[o];
        if (E > 0.0 && c.eps > 0.0 && (kx != 0.0 || ky != 0.0 || kz != 0.0)) I = E / ds_node;
        fI[o] = I;
        return;
    }
    const double ax = fx[o], ay = fy[o], az = fz[o];
    const double dn = sqrt(ax * ax + ay * ay + az * az);
    double kx = 0.0, ky = 0.0, kz = 0.0;
    if (c.eps > 0.0 && dn > 0.0) {
        if (E > 0.0) I = E / ds_node;
        kx = kmag * (ax / dn);
        ky = kmag * (ay / dn);
        kz = kmag * (az / dn);
    }
    fI[o] = I; fx[o] = kx; fy[o] = ky; fz[o] = kz;
}

// fields (E, Dx, Dy, Dz) -> gain coefficient, one wavefront per 2 x 4 x 8 brick of deposit-grid cells
// (z fastest: every load is eight 64-B runs).  A compact brick keeps the set of beams present
// ANYWHERE in the wave small -- the beam loops below run over that set (a ballot-built bit mask),
// and a 64-cell z-row crosses several times more beams than a brick does.
//   phase 1: every entry a beam's rays touched (E != 0) is normalised in place to (I, kx, ky, kz), I = 0 where the
//            beam is not present (E <= 0); with GainArgs.frozen only the energy entry (normalise_entry).
//   phase 2: K_i = sum_{j != i} G_ij I_j, beams in increasing order; gain <- gain + relax (K - gain),
//            stored only where it changes.
__global__ void __launch_bounds__(256) k_gain_field(const GainArgs a)
{
    const int HY = a.ny + 2, HZ = a.nz + 2;
    const long hsize = a.bstride;                 // entries stored per beam (the whole haloed grid, or one x-slab of it)
    const long total = hsize * a.nbeams;
    const int bx0 = a.hx_lo >> 1, bx = ((a.hx_hi + 1) >> 1) - bx0;   // brick columns touching the slab [hx_lo, hx_hi)
    const int by = (HY + 3) / 4, bz = (HZ + 7) / 8;
    const long bricks = (long)bx * by * bz;
    const int lane = threadIdx.x & (kWave - 1);
    const long wave0 = (long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const long nwaves = (long)gridDim.x * (blockDim.x / kWave);
    const double iaw2 = a.iaw * a.iaw;
    double sum_change = 0.0, sum_abs = 0.0;
    for (long brick = wave0; brick < bricks; brick += nwaves) {
        const int ibz = (int)(brick % bz);
        const long t = brick / bz;
        const int iby = (int)(t % by), ibx = bx0 + (int)(t / by);
        const int hi = 2 * ibx + (lane >> 5), hj = 4 * iby + ((lane >> 3) & 3), hk = 8 * ibz + (lane & 7);
        const bool valid = hi >= a.hx_lo && hi < a.hx_hi && hj < HY && hk < HZ;
        const long h = valid ? ((long)hi * HY + hj) * HZ + hk : a.store0;
        const long hs = h - a.store0;               // index into the (possibly slab-packed) arrays
        double *fI = a.fields + hs, *fx = fI + total, *fy = fx + total, *fz = fy + total;
        const CellState c = cell_state(a, h);
        const double kmag = a.k0 * c.rt;
        const double ds_node = (kC * c.rt) * a.dt;  // group speed x dt: energy x length -> intensity
        unsigned long long mask = 0ull;             // beams present in some cell of this brick
        for (int b = 0; b < a.nbeams; ++b) {
            const long o = (long)b * hsize;
            const double E = valid ? fI[o] : 0.0;
            if (__builtin_amdgcn_ballot_w64(E != 0.0) == 0ull) continue;   // no ray of this beam came near the brick
            if (__builtin_amdgcn_ballot_w64(E > 0.0) != 0ull) mask |= 1ull << b;
            if (E != 0.0) normalise_entry(a, c, kmag, ds_node, E, fI, fx, fy, fz, o);
        }
        const double pref = c.eps > 0.0 ? a.gain_const * c.frac * (1.0 / a.iaw) / c.rt : 0.0;
        for (int bi = 0; bi < a.nbeams; ++bi) {
            const long oi = (long)bi * hsize;
            double raw = 0.0;
            if ((mask >> bi) & 1ull) {
                const double Ii = fI[oi];
                const double kxi = fx[oi], kyi = fy[oi], kzi = fz[oi];
                double acc = 0.0;
                unsigned long long m = mask & ~(1ull << bi);
                while (m != 0ull) {
                    const int bj = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const long oj = (long)bj * hsize;
                    const double Ij = fI[oj];
                    const double qx = fx[oj] - kxi, qy = fy[oj] - kyi, qz = fz[oj] - kzi;
                    const double kiaw = sqrt(qx * qx + qy * qy + qz * qz);
                    if (valid && Ii > 0.0 && Ij > 0.0 && kiaw > 0.0) {
                        const double eta = (0.0 - (qx * c.ux + qy * c.uy + qz * c.uz)) / (kiaw * a.cs + 1e-10);
                        const double e2 = eta * eta;
                        const double P = iaw2 * eta / ((e2 - 1.0) * (e2 - 1.0) + iaw2 * e2);
                        acc += pref * P * Ij;
                    }
                }
                raw = acc;
            }
            if (valid) {
                double *gp = a.gain + oi + hs;
                const double old = *gp;
                const double nw = old + a.relax * (raw - old);
                if (nw != old) *gp = nw;
                sum_change += fabs(nw - old);
                sum_abs += fabs(nw);
            }
        }
    }
    if (a.change) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum_change += __shfl_xor(sum_change, off, kWave);
            sum_abs += __shfl_xor(sum_abs, off, kWave);
        }
        if (lane == 0) {
            atomicAdd(&a.change[0], sum_change);
            atomicAdd(&a.change[1], sum_abs);
        }
    }
}


// The same update with every unordered beam pair evaluated ONCE (G_ji = -G_ij exactly: eta changes sign,
// P is odd): the present beams of a brick are taken in tiles of GT, a tile pair (A, B) is GT x GT
// statically unrolled pair bodies on registers, A's sums stay in registers across its B tiles and B's go
// to `scratch` (read-modify-write by the owning lane; the lines stay in L1/L2).  Half the pair
// evaluations of k_gain_field and an eighth of its loads; sums are grouped by tile, so K differs
// from the ordered kernel's in the last bits only.
// What holds it (round 3, rocprofv3 --pmc at 256^3 / 60 beams, profiles/r3/cbet/pmc_summary.txt): 108 GB fetched
// (2 x FETCH_SIZE) + 14 GB written per call against ~50 GB of compulsory traffic = 5.9 TB/s over 20.6 ms, 79 % of the
// wave cycles waiting, SQ_INSTS_VALU 3.0e9 = 24 % of the vector issue rate: the re-streamed B tiles of the ~20 beams
// present per cell miss L1 and L2 (a brick's entries are 41 KB, sixteen wavefronts per CU hold sixteen bricks).  Two
// restructurings that read every entry once were built and measured slower: an 8-cell brick per wavefront staged in
// LDS with the pairs dealt to 8 lanes per cell (27.6 ms: 8 wavefronts per CU, one LDS atomic pair per evaluation) and
// one 64-cell brick per workgroup with the A tiles dealt to its four wavefronts and the sums in LDS (26.6 ms:
// imbalance across the wavefronts, two barriers per brick) -- profiles/r3/experiments/timing_variants.log.  Occupancy: 162
// registers give 12 wavefronts per CU; capped at 8 by a dummy LDS allocation 24.6 ms, at 4: 38.5 ms; forced to 128 registers
// (16 wavefronts, 20 spilled) 20.2 ms -- it wants wavefronts in flight as much as it wants bytes.
constexpr int GT = 4;   // measured at 256^3, 60 beams: 2 -> 27.8 ms, 3 -> 22.5, 4 -> 20.8, 5 -> 23.2, 6 -> 22.7, 8 -> 34.8 (register pressure)

struct BeamAtCell {
    double I, kx, ky, kz;
};

// pref * P(eta_ij) for the pair (i, j); zero intensities make the products vanish, no branch needed
__device__ __forceinline__ double pair_gain(const BeamAtCell &bi, const BeamAtCell &bj, double ux, double uy, double uz,
                                            double cs, double iaw2, double pref)
{
    const double qx = bj.kx - bi.kx, qy = bj.ky - bi.ky, qz = bj.kz - bi.kz;
    const double kiaw = sqrt(qx * qx + qy * qy + qz * qz);
    const double eta = (0.0 - (qx * ux + qy * uy + qz * uz)) / (kiaw * cs + 1e-10);
    const double e2 = eta * eta;
    const double P = iaw2 * eta / ((e2 - 1.0) * (e2 - 1.0) + iaw2 * e2);
    return pref * P;
}

__global__ void __launch_bounds__(256) k_gain_field_sym(const GainArgs a)
{
    const int HY = a.ny + 2, HZ = a.nz + 2;
    const long hsize = a.bstride;                 // entries stored per beam (the whole haloed grid, or one x-slab of it)
    const long total = hsize * a.nbeams;
    const int bx0 = a.hx_lo >> 1, bx = ((a.hx_hi + 1) >> 1) - bx0;   // brick columns touching the slab [hx_lo, hx_hi)
    const int by = (HY + 3) / 4, bz = (HZ + 7) / 8;
    const long bricks = (long)bx * by * bz;
    const int lane = threadIdx.x & (kWave - 1);
    const long wave0 = (long)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const long nwaves = (long)gridDim.x * (blockDim.x / kWave);
    const double iaw2 = a.iaw * a.iaw;
    double sum_change = 0.0, sum_abs = 0.0;
    for (long brick = wave0; brick < bricks; brick += nwaves) {
        const int ibz = (int)(brick % bz);
        const long t = brick / bz;
        const int iby = (int)(t % by), ibx = bx0 + (int)(t / by);
        const int hi = 2 * ibx + (lane >> 5), hj = 4 * iby + ((lane >> 3) & 3), hk = 8 * ibz + (lane & 7);
        const bool valid = hi >= a.hx_lo && hi < a.hx_hi && hj < HY && hk < HZ;
        const long h = valid ? ((long)hi * HY + hj) * HZ + hk : a.store0;
        const long hs = h - a.store0;               // index into the (possibly slab-packed) arrays
        double *fI = a.fields + hs, *fx = fI + total, *fy = fx + total, *fz = fy + total;
        double *raw = a.scratch + hs;
        const CellState c = cell_state(a, h);
        const double kmag = a.k0 * c.rt;
        const double ds_node = (kC * c.rt) * a.dt;
        unsigned long long mask = 0ull;
        for (int b = 0; b < a.nbeams; ++b) {  // phase 1: as k_gain_field, plus the scratch sums start at zero
            const long o = (long)b * hsize;
            const double E = valid ? fI[o] : 0.0;
            if (__builtin_amdgcn_ballot_w64(E != 0.0) == 0ull) continue;   // no ray of this beam came near the brick
            if (E != 0.0) normalise_entry(a, c, kmag, ds_node, E, fI, fx, fy, fz, o);   // also clears a non-positive E
            if (__builtin_amdgcn_ballot_w64(E > 0.0) == 0ull) continue;
            mask |= 1ull << b;
            if (valid) raw[o] = 0.0;
        }
        const double pref = (valid && c.eps > 0.0) ? a.gain_const * c.frac * (1.0 / a.iaw) / c.rt : 0.0;
        auto load_tile = [&](unsigned long long &m, int (&id)[GT], BeamAtCell (&bm)[GT]) {
#pragma unroll
            for (int s = 0; s < GT; ++s) {
                id[s] = -1;
                bm[s].I = 0.0; bm[s].kx = 0.0; bm[s].ky = 0.0; bm[s].kz = 0.0;
                if (m != 0ull) {
                    id[s] = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const long o = (long)id[s] * hsize;
                    const double I = fI[o];
                    bm[s].I = I > 0.0 ? I : 0.0;          // an absent beam's entry is whatever was deposited: mask it
                    bm[s].kx = fx[o]; bm[s].ky = fy[o]; bm[s].kz = fz[o];
                }
            }
        };
        unsigned long long ma = mask;
        while (ma != 0ull) {
            int aid[GT];
            BeamAtCell A[GT];
            load_tile(ma, aid, A);
            double KA[GT];
#pragma unroll
            for (int s = 0; s < GT; ++s) KA[s] = 0.0;
#pragma unroll
            for (int s = 0; s < GT; ++s)
#pragma unroll
                for (int u = s + 1; u < GT; ++u) {
                    const double g = pair_gain(A[s], A[u], c.ux, c.uy, c.uz, a.cs, iaw2, pref);
                    KA[s] += g * A[u].I;
                    KA[u] -= g * A[s].I;
                }
            unsigned long long mb = ma;
            while (mb != 0ull) {
                int bid[GT];
                BeamAtCell B[GT];
                load_tile(mb, bid, B);
                double KB[GT];
#pragma unroll
                for (int u = 0; u < GT; ++u) KB[u] = 0.0;
#pragma unroll
                for (int s = 0; s < GT; ++s)
#pragma unroll
                    for (int u = 0; u < GT; ++u) {
                        const double g = pair_gain(A[s], B[u], c.ux, c.uy, c.uz, a.cs, iaw2, pref);
                        KA[s] += g * B[u].I;
                        KB[u] -= g * A[s].I;
                    }
#pragma unroll
                for (int u = 0; u < GT; ++u)
                    if (bid[u] >= 0 && valid) raw[(long)bid[u] * hsize] += KB[u];
            }
#pragma unroll
            for (int s = 0; s < GT; ++s)
                if (aid[s] >= 0 && valid) {
                    const long o = (long)aid[s] * hsize;
                    // a beam that is absent from THIS cell has K = 0 (its sums above came from whatever its
                    // entry held); its intensity was masked to zero, so it gave nothing to the others
                    const double r = A[s].I > 0.0 ? raw[o] + KA[s] : 0.0;
                    double *gp = a.gain + o + hs;
                    const double old = *gp;
                    const double nw = old + a.relax * (r - old);
                    if (nw != old) *gp = nw;
                    sum_change += fabs(nw - old);
                    sum_abs += fabs(nw);
                }
        }
        if (a.consume && valid) {
            // consume: the energy entries have done their work -- hand them back zeroed, so that the next field pass
            // can accumulate into them without a memset in between (the lines were just written, they are in L2);
            // the direction entries stay: later passes reuse them (GainArgs.frozen)
            unsigned long long mz = mask;
            while (mz != 0ull) {
                const long o = (long)(__ffsll((long long)mz) - 1) * hsize;
                mz &= mz - 1;
                fI[o] = 0.0;
            }
        }
        if (valid) {  // beams absent from the whole brick relax towards zero
            unsigned long long rest = ~mask & (a.nbeams >= 64 ? ~0ull : ((1ull << a.nbeams) - 1));
            while (rest != 0ull) {
                const int b = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                double *gp = a.gain + (long)b * hsize + hs;
                const double old = *gp;
                const double nw = old + a.relax * (0.0 - old);
                if (nw != old) *gp = nw;
                sum_change += fabs(nw - old);
                sum_abs += fabs(nw);
            }
        }
    }
    if (a.change) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum_change += __shfl_xor(sum_change, off, kWave);
            sum_abs += __shfl_xor(sum_abs, off, kWave);
        }
        if (lane == 0) {
            atomicAdd(&a.change[0], sum_change);
            atomicAdd(&a.change[1], sum_abs);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// main.cu:334-349 (commented out there): edepavg[i][j][k] = (27 haloed cells around node (i,j,k)) / 27,
// terms added in the order the reference writes them (i offset fastest, then j, then k).  A 4 x 4 x 64
// output tile per workgroup, its 6 x 6 x 66 input block staged through LDS: HBM sees every input once.
// ---------------------------------------------------------------------------------------------
constexpr int kAvgTI = 4, kAvgTJ = 4, kAvgTK = 64, kAvgPad = kAvgTK + 3;

__global__ void __launch_bounds__(256) k_edep_average(const double *__restrict__ edep, double *__restrict__ out,
                                                      int nx, int ny, int nz)
{
    __shared__ double t[kAvgTI + 2][kAvgTJ + 2][kAvgPad];
    const int tk = (nz + kAvgTK - 1) / kAvgTK, tj = (ny + kAvgTJ - 1) / kAvgTJ;
    const int bk = blockIdx.x % tk, bj = (blockIdx.x / tk) % tj, bi = blockIdx.x / (tk * tj);
    const int i0 = bi * kAvgTI, j0 = bj * kAvgTJ, k0 = bk * kAvgTK;
    const long sY = nz + 2, sX = (long)(ny + 2) * (nz + 2);
    for (int idx = threadIdx.x; idx < (kAvgTI + 2) * (kAvgTJ + 2) * (kAvgTK + 2); idx += blockDim.x) {
        const int c = idx % (kAvgTK + 2), b = (idx / (kAvgTK + 2)) % (kAvgTJ + 2), a = idx / ((kAvgTK + 2) * (kAvgTJ + 2));
        const int gi = i0 + a, gj = j0 + b, gk = k0 + c;
        t[a][b][c] = (gi < nx + 2 && gj < ny + 2 && gk < nz + 2) ? edep[gi * sX + gj * sY + gk] : 0.0;
    }
    __syncthreads();
    const int lk = threadIdx.x & (kAvgTK - 1), lj = threadIdx.x / kAvgTK;
    const int j = j0 + lj, k = k0 + lk;
    if (j >= ny || k >= nz) return;
#pragma unroll
    for (int li = 0; li < kAvgTI; ++li) {
        const int i = i0 + li;
        if (i >= nx) break;
        double acc = t[li][lj][lk];
#pragma unroll
        for (int dk = 0; dk < 3; ++dk)
#pragma unroll
            for (int dj = 0; dj < 3; ++dj)
#pragma unroll
                for (int di = 0; di < 3; ++di)
                    if (dk + dj + di != 0) acc = acc + t[li + di][lj + dj][lk + dk];
        out[((long)i * ny + j) * nz + k] = acc / 27;
    }
}

// ---------------------------------------------------------------------------------------------
// Sparse exchange of the slab-owned CBET loop (tracer._Exchanger): a beam's rays touch ~10 % of a slab, so what a rank
// sends a slab owner is the list of 64-byte z-runs ("segments": 8 doubles, aligned to 8 along z) its beams can ever
// deposit into, not the dense sub-array.  A segment is {beam (row of the array), index of the run inside one beam's
// [planes][ny+2][ceil((nz+2)/8)] run grid}; the lists are fixed for the life of a solve (ray paths do not depend on
// the gain) and live in device memory.  pack gathers the runs of one message into a contiguous buffer (64-B stores),
// unpack scatters a received buffer; the run that straddles the end of a z-row is zero-filled / clipped.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pack_segments(const double *__restrict__ src, long beam_stride, int hz, int zsegs,
                                                        const int2 *__restrict__ seg, long nseg, double *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one double per thread: 8 threads = one 64-B run
    if (i >= nseg * 8) return;
    const int2 sg = seg[i >> 3];
    const int k = (int)(i & 7), zs = sg.y % zsegs, row = sg.y / zsegs;   // row = plane * (ny + 2) + y
    const int z = 8 * zs + k;
    out[i] = z < hz ? src[(long)sg.x * beam_stride + (long)row * hz + z] : 0.0;
}

__global__ void __launch_bounds__(256) k_unpack_segments(double *__restrict__ dst, long beam_stride, int hz, int zsegs,
                                                          const int2 *__restrict__ seg, long nseg, const double *__restrict__ in)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * 8) return;
    const int2 sg = seg[i >> 3];
    const int k = (int)(i & 7), zs = sg.y % zsegs, row = sg.y / zsegs;
    const int z = 8 * zs + k;
    if (z < hz) dst[(long)sg.x * beam_stride + (long)row * hz + z] = in[i];
}

}  // namespace

hipError_t launch_pack_segments(const double *src, long beam_stride, int hy, int hz, const int *seg, long nseg, double *out,
                                hipStream_t stream)
{
    if (nseg <= 0) return hipSuccess;
    const long blocks = (nseg * 8 + 255) / 256;
    (void)hy;
    hipLaunchKernelGGL(k_pack_segments, dim3((unsigned)blocks), dim3(256), 0, stream, src, beam_stride, hz, (hz + 7) / 8,
                       reinterpret_cast<const int2 *>(seg), nseg, out);
    return hipGetLastError();
}

hipError_t launch_unpack_segments(double *dst, long beam_stride, int hy, int hz, const int *seg, long nseg, const double *in,
                                  hipStream_t stream)
{
    if (nseg <= 0) return hipSuccess;
    const long blocks = (nseg * 8 + 255) / 256;
    (void)hy;
    hipLaunchKernelGGL(k_unpack_segments, dim3((unsigned)blocks), dim3(256), 0, stream, dst, beam_stride, hz, (hz + 7) / 8,
                       reinterpret_cast<const int2 *>(seg), nseg, in);
    return hipGetLastError();
}

hipError_t launch_edep_average(const double *edep, double *out, int nx, int ny, int nz, hipStream_t stream)
{
    const long blocks = (long)((nx + kAvgTI - 1) / kAvgTI) * ((ny + kAvgTJ - 1) / kAvgTJ) * ((nz + kAvgTK - 1) / kAvgTK);
    hipLaunchKernelGGL(k_edep_average, dim3((unsigned)blocks), dim3(256), 0, stream, edep, out, nx, ny, nz);
    return hipGetLastError();
}

hipError_t launch_gain_field(const GainArgs &a, hipStream_t stream)
{
    if (a.hx_hi <= a.hx_lo) return hipSuccess;
    const long bricks = (long)(((a.hx_hi + 1) >> 1) - (a.hx_lo >> 1)) * ((a.ny + 5) / 4) * ((a.nz + 9) / 8);  // 2 x 4 x 8 cells of the haloed grid each
    long blocks = (bricks + 3) / 4;                                                    // four wavefronts per workgroup
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (a.scratch) hipLaunchKernelGGL(k_gain_field_sym, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_gain_field, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cbet
