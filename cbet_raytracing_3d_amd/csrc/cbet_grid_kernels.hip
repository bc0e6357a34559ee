// cbet_grid_kernels.hip -- the cell-parallel kernels beside the ray integrator (cbet_kernels.hip):
//   * k_gain_field (ordered), k_gain_field_sym (pairs once, LDS-staged) : per-beam fields -> CBET gain coefficient (no
//     reference counterpart, parity unpinned -- model and layout in DESIGN.md section 9)
//   * k_edep_average                 : the 27-point node average of main.cu:334-349
// Built with -ffp-contract=off like the rest of the library: every statement of the ordered kernel and of the
// normalisation is one IEEE operation in the order written, which is the order the CPU checker of the CBET stage
// uses; the pair-once kernel's pair function (pair_gain_fast) is the exception and says so.
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


struct BeamAtCell {
    double I, kx, ky, kz;
};

// ---------------------------------------------------------------------------------------------
// The same update with every unordered beam pair evaluated ONCE (G_ji = -G_ij exactly: eta changes sign, P is odd) and
// every entry read from memory ONCE: one wavefront (= one workgroup) per run of LC = 16 cells along z, LG = 4 lanes per
// cell, the run's present beams staged in LDS.
//   masks  : lane (cell, g) loads beam 4 r + g in round r, all rounds in flight; the values are transposed through LDS
//            so that lane b holds beam b's sixteen entries and one compare per cell accumulates the masks of the beams
//            that touched the run (E != 0 somewhere) and that are present in it (E > 0 somewhere) in scalar registers.
//   phase 1: the touched beams, four per round: normalise the entry as k_gain_field does, write it back, and stage
//            (I, kx, ky, kz) of the present beams in LDS, compacted to slots, with a zeroed sum beside them.
//   phase 2: the slots in tiles of four.  Inside a tile lane group g pairs its own beam with the next and (g < 2) the
//            next but one, cyclically -- six pairs in two evaluations; against the later tiles the A tile sits in the
//            registers of all four lanes of the cell and lane group g takes slot g of every B tile, so its sum for that
//            B beam is complete (no atomics) and the four partial sums of the A beams are folded across the lane groups
//            once per A tile.
//   phase 3: every beam's gain relaxes towards its sum (zero where the beam is absent from the cell).
// A run crossed by more beams than LDS has slots for (LCAP = 20) is taken in halves or quarters: fewer cells, more slots.
// LDS 5 arrays x LCAP x LC doubles = 12.8 KB and 168 registers: twelve wavefronts per CU.  The plasma state of a cell
// (cell_state: two square roots, five divisions) is computed for 64 cells at a time and handed out by lane shuffles.
// The sums live in LDS: GainArgs.scratch only selects this kernel, it is never dereferenced.
//
// Measured (round 3, 256^3, 60 beams, frozen directions; profiles/r3/cbet/): 10.9-11.4 ms against 19.2 ms for the kernel
// it replaces (one wavefront per 2x4x8 brick, B tiles re-streamed from memory, sums in a scratch array: 108 GB fetched +
// 14 GB written per call, 70 % of its 1.3e9 line requests missing L2); now ~49 GB fetched + 5.2 GB written,
// SQ_INSTS_VALU 3.7e9 = 0.55 of the vector issue rate.  The steps in between (profiles/r3/experiments/gain_kernel.log):
// 64 slots and 4 waves per CU 27.9 ms; 32 slots / 8 waves with the exact IEEE pair function 22.8; the pair function as
// one quotient, in-tile pairs in two evaluations, phase 1 over touched beams only 14.4 (20 slots: 12.5); masks by
// transposition instead of ballots + 530 scalar bit tests per run, plasma state per 64 cells 11.8; all of phase 1's
// loads in flight 11.4.  LCAP 16 / 24 / 28 / 32: 13.0 / 12.0 / 13.2 / 13.4 ms; LDS-DMA prefetch of the next run's
// lines 13.4 ms; FROZEN as a template parameter 11.5 -> 11.1; skipping phase-3 rounds whose four beams are absent and
// carry no gain: nothing.  Where the 11.2 ms go (timing builds that skip phases): masks 2.3, phase 1 2.5, phase 2 4.0,
// phase 3 2.6 -- additive: a wavefront's phases do not overlap with each other, only with other wavefronts'.  Whole-line
// z-rows (grids with nz + 2 a multiple of 16: 270^3) are ~8 % faster per cell, not more: the memory phases wait on
// latency at 12 waves per CU, not on bytes.
// ---------------------------------------------------------------------------------------------
constexpr int LC = 16, LG = 4, LCAP = 20, LROUNDS = 64 / LG;
constexpr int LCH = 4, LCH3 = 8;   // rounds whose loads are in flight together in phase 1 / phase 3

// pref * P(eta_ij) as ONE quotient: with N = -(q . u) and D = |q| cs + 1e-10 (eta = N / D),
//   P = iaw^2 eta / ((eta^2 - 1)^2 + iaw^2 eta^2) = iaw^2 N D^3 / ((N^2 - D^2)^2 + iaw^2 N^2 D^2),
// square root and reciprocal by the hardware estimates and Newton steps (a few ulp; the sums of the symmetric kernels are
// grouped differently from the ordered kernel's anyway).  q = 0 (a beam against itself, or two parallel beams) gives
// N = 0 over D^4 = 1e-40: zero, as in the ordered kernel.
__device__ __forceinline__ double pair_gain_fast(const BeamAtCell &bi, const BeamAtCell &bj, double ux, double uy, double uz,
                                                 double cs, double iaw2, double pref_iaw2)
{
    const double qx = bj.kx - bi.kx, qy = bj.ky - bi.ky, qz = bj.kz - bi.kz;
    const double q2 = fmax(__builtin_fma(qz, qz, __builtin_fma(qy, qy, qx * qx)), 1e-280);
    // |q| = sqrt(q2): one Goldschmidt step from the reciprocal-square-root estimate, one correction
    const double r0 = __builtin_amdgcn_rsq(q2);
    double gq = q2 * r0, hq = 0.5 * r0;
    const double e = __builtin_fma(-hq, gq, 0.5);
    gq = __builtin_fma(gq, e, gq); hq = __builtin_fma(hq, e, hq);
    gq = __builtin_fma(__builtin_fma(-gq, gq, q2), hq, gq);
    const double N = -__builtin_fma(qz, uz, __builtin_fma(qy, uy, qx * ux));
    const double D = __builtin_fma(gq, cs, 1e-10);
    const double N2 = N * N, D2 = D * D, t = N2 - D2;
    const double den = __builtin_fma(t, t, (iaw2 * N2) * D2);
    const double num = ((pref_iaw2 * N) * D) * D2;
    // num / den: reciprocal estimate, one Newton step, one correction of the quotient
    double y = __builtin_amdgcn_rcp(den);
    y = __builtin_fma(__builtin_fma(-den, y, 1.0), y, y);
    const double qt = num * y;
    return __builtin_fma(__builtin_fma(-den, qt, num), y, qt);
}

// LDS traffic between lanes of ONE wavefront: the hardware keeps a wavefront's LDS instructions in order, the compiler
// has to be told that they may not be reordered across this point
__device__ __forceinline__ void lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool FROZEN>
__global__ void __launch_bounds__(64, 3) k_gain_field_sym(const GainArgs a)
{
    static_assert(5 * LCAP * LC >= 64 * (LC + 1), "the slot arrays double as the staging area of the presence masks");
    __shared__ double lds[5 * LCAP * LC];
    double *const sI = lds, *const sX = sI + LCAP * LC, *const sY = sX + LCAP * LC, *const sZ = sY + LCAP * LC,
                 *const sR = sZ + LCAP * LC;
    const int HY = a.ny + 2, HZ = a.nz + 2;
    const long hsize = a.bstride;
    const long total = hsize * a.nbeams;
    const long rows = (long)(a.hx_hi - a.hx_lo) * HY;
    const int lane = threadIdx.x, c = lane & (LC - 1), g = lane >> 4;
    const double iaw2 = a.iaw * a.iaw;
    const int rounds = (a.nbeams + LG - 1) / LG;
    double sum_change = 0.0, sum_abs = 0.0;
    double st_rt = 0.0, st_ux = 0.0, st_uy = 0.0, st_uz = 0.0, st_pref = 0.0;
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const int hi = a.hx_lo + (int)(row / HY), hj = (int)(row % HY);
        // runs start on 128-byte lines of the arrays, not at z = 16 k of the row: the row's first entry sits `sft` doubles into
        // a line (rows are nz + 2 doubles long), so the first run holds the 16 - sft cells up to the next line and every later
        // one is exactly one line per beam and array (beams whose stride from beam 0 is an odd multiple of 8 doubles -- every
        // second one at 256^3 -- stay half a line off: the arrays' beam stride is the caller's)
        const int sft = (int)((((long)hi * HY + hj) * HZ - a.store0) & (LC - 1));
        const int zb_row = (HZ + sft + LC - 1) / LC;
        for (int ibz = 0; ibz < zb_row; ++ibz) {
            const int hk = LC * ibz + c - sft;
            const bool valid = hk >= 0 && hk < HZ;
            const long h = ((long)hi * HY + hj) * HZ + (valid ? hk : 0);
            const long hs = h - a.store0;
            double *fI = a.fields + hs, *fx = fI + total, *fy = fx + total, *fz = fy + total;
            // ---- which beams touched the run (E != 0 somewhere), which are present (E > 0 somewhere): every round's
            //      load in flight together, transposed through LDS so that lane b sees beam b's sixteen entries and one
            //      compare per cell accumulates the masks (bit b = lane b) in scalar registers
            unsigned long long mask = 0ull, touched = 0ull;
            {
                double E[LROUNDS];
#pragma unroll
                for (int r = 0; r < LROUNDS; ++r) {
                    const int b = LG * r + g;
                    E[r] = (r < rounds && b < a.nbeams && valid) ? fI[(long)b * hsize] : 0.0;
                }
                __syncthreads();                 // the previous run's LDS reads are done
#pragma unroll
                for (int r = 0; r < LROUNDS; ++r) lds[(LG * r + g) * (LC + 1) + c] = E[r];
                __syncthreads();
#pragma unroll
                for (int k = 0; k < LC; ++k) {
                    const double e = lds[lane * (LC + 1) + k];
                    mask |= __builtin_amdgcn_ballot_w64(e > 0.0);
                    touched |= __builtin_amdgcn_ballot_w64(e != 0.0);
                }
            }
            const int n = __popcll(mask), tiles = (n + LG - 1) / LG;
            // the plasma state of 64 cells at a time (lane = cell), handed to the four runs they make up
            if ((ibz & 3) == 0) {
                const int hk64 = LC * ibz + lane - sft;
                const CellState c64 = cell_state(a, ((long)hi * HY + hj) * HZ + (hk64 < 0 ? 0 : (hk64 < HZ ? hk64 : HZ - 1)));
                st_rt = c64.rt; st_ux = c64.ux; st_uy = c64.uy; st_uz = c64.uz;
                st_pref = c64.eps > 0.0 ? a.gain_const * c64.frac * (1.0 / a.iaw) / c64.rt : 0.0;
            }
            const int from = LC * (ibz & 3) + c;
            const double rt = __shfl(st_rt, from, kWave), ux = __shfl(st_ux, from, kWave), uy = __shfl(st_uy, from, kWave),
                         uz = __shfl(st_uz, from, kWave);
            const double pref = valid ? __shfl(st_pref, from, kWave) : 0.0;
            const bool subcritical = rt > 0.0;   // eps > 0
            const double kmag = a.k0 * rt;
            const double ds_node = (kC * rt) * a.dt;
            const double pref_iaw2 = pref * iaw2;
            // a run crossed by more beams than LDS has slots for is taken in halves or quarters (fewer cells, more slots)
            const int sh = n <= LCAP ? 4 : (n <= 2 * LCAP ? 3 : 2), lc = 1 << sh, cl = c & (lc - 1);
            for (int part = 0; part < (LC >> sh); ++part) {
                const bool mine_ = (c >> sh) == part;
                __syncthreads();                 // the previous part's LDS reads are done
                // ---- phase 1: the touched beams, four per round (lane group g takes the g-th of them): normalise the
                //      entry (the energy is in L1 / L2 from the loads above), write it back, stage it if the beam is present
                if (mine_) {
                    long hs1 = hsize;            // opaque: keeps the address arithmetic of this phase out of the pair loop's registers
                    asm volatile("" : "+s"(hs1));
                    unsigned long long tm = touched;
                    while (tm != 0ull) {
                        int bb[LCH];
                        double E[LCH], X[LCH], Y[LCH], Z[LCH];
#pragma unroll
                        for (int u = 0; u < LCH; ++u) {
                            int bq[LG];
#pragma unroll
                            for (int q = 0; q < LG; ++q) {
                                bq[q] = tm != 0ull ? __ffsll((long long)tm) - 1 : -1;
                                tm &= tm - 1ull;
                            }
                            bb[u] = g == 0 ? bq[0] : (g == 1 ? bq[1] : (g == 2 ? bq[2] : bq[3]));
                            // the three direction entries are fetched whether this cell's energy entry turns out to be
                            // zero or not (few are): all four loads of all rounds of the chunk are in flight together
                            const bool in = bb[u] >= 0 && valid;
                            const long o = (long)bb[u] * hs1;
                            E[u] = in ? fI[o] : 0.0;
                            X[u] = in ? fx[o] : 0.0;
                            Y[u] = in ? fy[o] : 0.0;
                            Z[u] = in ? fz[o] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < LCH; ++u) {
                            const int b = bb[u];
                            const long o = (long)b * hs1;
                            double I = 0.0;
                            if (E[u] != 0.0) {
                                if (FROZEN) {
                                    if (E[u] > 0.0 && subcritical && (X[u] != 0.0 || Y[u] != 0.0 || Z[u] != 0.0)) I = E[u] / ds_node;
                                } else {
                                    const double ax = X[u], ay = Y[u], az = Z[u];
                                    const double dn = sqrt(ax * ax + ay * ay + az * az);
                                    X[u] = Y[u] = Z[u] = 0.0;
                                    if (subcritical && dn > 0.0) {
                                        if (E[u] > 0.0) I = E[u] / ds_node;
                                        X[u] = kmag * (ax / dn);
                                        Y[u] = kmag * (ay / dn);
                                        Z[u] = kmag * (az / dn);
                                    }
                                    fx[o] = X[u]; fy[o] = Y[u]; fz[o] = Z[u];
                                }
                                fI[o] = a.consume ? 0.0 : I;
                            }
                            if (b >= 0 && ((mask >> b) & 1ull)) {
                                const int at = (__popcll(mask & ((1ull << b) - 1ull)) << sh) + cl;
                                const bool here = I > 0.0;   // a beam absent from THIS cell gives and takes nothing
                                sI[at] = here ? I : 0.0;
                                sX[at] = here ? X[u] : 0.0;
                                sY[at] = here ? Y[u] : 0.0;
                                sZ[at] = here ? Z[u] : 0.0;
                                sR[at] = 0.0;
                            }
                        }
                    }
                    if (n + g < LG * tiles) {    // the padding slots of the last tile
                        const int at = ((n + g) << sh) + cl;
                        sI[at] = 0.0; sX[at] = 0.0; sY[at] = 0.0; sZ[at] = 0.0; sR[at] = 0.0;
                    }
                }
                __syncthreads();
                // ---- phase 2: every unordered pair once
                if (mine_) {
                    for (int ta = 0; ta < tiles; ++ta) {
                        // inside the tile: lane group g pairs its own beam g with beams g+1 and (g < 2) g+2, cyclically --
                        // the six pairs in two evaluations; each lane adds to its own beam's sum and to the partner's,
                        // and the partners of one instruction are four different beams
                        {
                            const int self = ((LG * ta + g) << sh) + cl;
                            const int p1 = ((LG * ta + ((g + 1) & 3)) << sh) + cl, p2 = ((LG * ta + ((g + 2) & 3)) << sh) + cl;
                            BeamAtCell S, P1, P2;
                            S.I = sI[self]; S.kx = sX[self]; S.ky = sY[self]; S.kz = sZ[self];
                            P1.I = sI[p1]; P1.kx = sX[p1]; P1.ky = sY[p1]; P1.kz = sZ[p1];
                            P2.I = sI[p2]; P2.kx = sX[p2]; P2.ky = sY[p2]; P2.kz = sZ[p2];
                            const double g1 = pair_gain_fast(S, P1, ux, uy, uz, a.cs, iaw2, pref_iaw2);
                            const double g2 = g < 2 ? pair_gain_fast(S, P2, ux, uy, uz, a.cs, iaw2, pref_iaw2) : 0.0;
                            sR[self] += g1 * P1.I + g2 * P2.I;
                            lds_order();             // the next statement's entries are other lanes' `self`
                            sR[p1] -= g1 * S.I;
                            lds_order();
                            sR[p2] -= g2 * S.I;
                            lds_order();
                        }
                        if (ta + 1 == tiles) break;
                        BeamAtCell A[LG];
                        double KA[LG];
#pragma unroll
                        for (int s_ = 0; s_ < LG; ++s_) {
                            const int at = ((LG * ta + s_) << sh) + cl;
                            A[s_].I = sI[at]; A[s_].kx = sX[at]; A[s_].ky = sY[at]; A[s_].kz = sZ[at];
                            KA[s_] = 0.0;
                        }
                        for (int tb = ta + 1; tb < tiles; ++tb) {
                            const int bt = ((LG * tb + g) << sh) + cl;
                            BeamAtCell B;
                            B.I = sI[bt]; B.kx = sX[bt]; B.ky = sY[bt]; B.kz = sZ[bt];
                            double KB = 0.0;
#pragma unroll
                            for (int s_ = 0; s_ < LG; ++s_) {
                                const double gn = pair_gain_fast(A[s_], B, ux, uy, uz, a.cs, iaw2, pref_iaw2);
                                KA[s_] += gn * B.I;
                                KB -= gn * A[s_].I;
                            }
                            sR[bt] += KB;
                        }
                        // fold the four lane groups' partial sums of the A beams; group g keeps beam g's
                        double fold = 0.0;
#pragma unroll
                        for (int s_ = 0; s_ < LG; ++s_) {
                            double t = KA[s_];
                            t += __shfl_xor(t, 16, kWave);
                            t += __shfl_xor(t, 32, kWave);
                            if (s_ == g) fold = t;
                        }
                        sR[((LG * ta + g) << sh) + cl] += fold;
                    }
                }
                __syncthreads();
                // ---- phase 3: relax the gain of every beam towards its sum (zero where the beam is not in the cell)
                if (mine_ && valid) {
                    long hs3 = hsize;
                    asm volatile("" : "+s"(hs3));
                    for (int r0 = 0; r0 < rounds; r0 += LCH3) {
                        double old[LCH3];
#pragma unroll
                        for (int u = 0; u < LCH3; ++u) {
                            const int b = LG * (r0 + u) + g;
                            old[u] = b < a.nbeams ? a.gain[(long)b * hs3 + hs] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < LCH3; ++u) {
                            const int b = LG * (r0 + u) + g;
                            if (b >= a.nbeams) continue;
                            double raw = 0.0;
                            if ((mask >> b) & 1ull) {
                                const int at = (__popcll(mask & ((1ull << b) - 1ull)) << sh) + cl;
                                raw = sI[at] > 0.0 ? sR[at] : 0.0;
                            }
                            const double nw = old[u] + a.relax * (raw - old[u]);
                            if (nw != old[u]) a.gain[(long)b * hs3 + hs] = nw;
                            sum_change += fabs(nw - old[u]);
                            sum_abs += fabs(nw);
                        }
                    }
                }
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
    if (a.scratch) {                       // one single-wavefront workgroup per z-row of the slab
        const long rows = (long)(a.hx_hi - a.hx_lo) * (a.ny + 2);
        if (a.frozen) hipLaunchKernelGGL(k_gain_field_sym<true>, dim3((unsigned)rows), dim3(64), 0, stream, a);
        else hipLaunchKernelGGL(k_gain_field_sym<false>, dim3((unsigned)rows), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    const long bricks = (long)(((a.hx_hi + 1) >> 1) - (a.hx_lo >> 1)) * ((a.ny + 5) / 4) * ((a.nz + 9) / 8);  // 2 x 4 x 8 cells of the haloed grid each
    long blocks = (bricks + 3) / 4;                                                    // four wavefronts per workgroup
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_gain_field, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace cbet
