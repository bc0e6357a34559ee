// cbet_trace_window.hip -- the shipped ray integrator for gfx950 (CDNA4): CBET_KERNEL_LDS_WINDOW.
//
// One wavefront (64 lanes, one workgroup) = one ray bundle = one 8x8-ray patch of a beam's cross
// section.  Everything that decides where a ray goes and when it stops -- position, velocity, energy, cell --
// is the reference's arithmetic, operation for operation (/root/reference/launch_ray_XZ.cu:207-357; the
// file is built with -ffp-contract=off); the deposit is the reference's up to the last bits of its terms (14
// products for 20, see `accumulate`).  What is MI355X-specific is everything around it:
//
//   * plasma: one 32-byte record per node (k_step_table: the three kicks with the reference's edge rule baked in,
//     and the absorption coefficient), gathered once per step from inline assembly and waited for with a counted
//     vmcnt (record_issue / record_wait); the loop is rotated so that the flush of the pending sums, this step's
//     weights and the window logic all run between a gather and its wait.
//   * relocation: for cells deep inside the grid the reference's mutating-bound candidate loop
//     (:282-292) is a function of g = f - cell alone and every difference it forms is exact, so it is
//     evaluated with two comparisons (cbet_relocate.h, relocate_deep_interior; fuzzed against the
//     literal loop on the CPU).  Whether a wave is "deep inside" is decided on the SCALAR unit from the
//     deposit windows' origins; near the faces the wave takes the closed form with the face rules.
//   * deposit: wave-private dense LDS tiles of fp64 accumulators ("boxes") whose origins follow the
//     bundle.  A lane sums its ray's deposits in eight registers while the ray's eight target nodes stay the
//     same; when they change it issues eight ds_add_f64 into its box (plain trace; the CBET kernels
//     deposit every step).  The plane (or, along z, the 64-byte-aligned brick of 8 planes) that leaves a
//     box when its origin moves is written back with global fp64 atomics -- z-bricks make every such
//     atomic request a full 64-B line (the memory-side atomic path is priced per 64-B request,
//     MI355X_MICROARCH.md "Global float atomics").  A second box adopts lanes that leave the first
//     (bundles fan out after the turning point).  Corner order is lane-dependent so that rays sharing all
//     8 nodes hit different LDS addresses in any one ds_add_f64; the first box stores its rows densely
//     with a swizzled z index (Tile::zr), which spreads a bundle's nodes over the banks in 8 KB: with the second box
//     10,240 B per wave, 16 waves per CU.
//   * control: every lane predicate that steers the step is a 64-bit mask in scalar registers (live rays, box B's lanes,
//     lanes outside both boxes, who moved, who flushes), combined with scalar instructions and turned into a lane
//     condition only where lanes diverge (CBET_LANES); the loop's branches are all wave-uniform and the file is built
//     with -mllvm -structurizecfg-skip-uniform-regions=true, so they stay plain branches (structurised, their flow
//     blocks cost the common path ~40 scalar copies per step); one loop exit; the rare window arm works on copies and
//     writes back through moves tied to the variables' registers (commit).
//
// Template parameters: WZ = z extent of a tile (16: aligned z-bricks; 8: single z-planes, used by the
// CBET field pass whose three extra component tiles would not fit otherwise); GENERIC = run-time
// absorption flag and 64-bit table indexing (grids of >= 2^32 table bytes, bookkeeping mode) instead
// of the compiled-in common case; CBET = 0 none, 1 gain hooks, 2 gain hooks + the energy field deposited,
// 4 fused four-component field pass (no reference counterpart: DESIGN.md section 9); STATS = also count
// the deposit windows' diagnostics (cbet_params.window_stats).
#include <hip/hip_runtime.h>

#include <type_traits>

// The hand-counted vmcnt waits below assume ONE in-order vector-memory counter shared by loads, stores and atomics
// (gfx9 / CDNA) and the gfx950 instruction set: refuse any other device target.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "cbet_trace_window.hip is written for gfx950 (CDNA4) only"
#endif

#include "cbet_trace_common.h"

namespace cbet {
namespace {

#define CBET_BALLOT(cond) __builtin_amdgcn_ballot_w64(cond)

// A wave-private tile of fp64 accumulators covering WX x WY x WZ nodes (powers of two), addressed toroidally.
template <int WX_, int WY_, int WZ_, bool PAD, bool XORSW = false>
struct Tile {
    static constexpr int WX = WX_, WY = WY_, WZ = WZ_;
    static constexpr int XM = WX - 1, YM = WY - 1, ZM = WZ - 1;
    // Layout (doubles).  One ds_add_f64 costs the CU 8.3 cycles + 2 per extra lane on the busiest bank of each 16-lane
    // group (bank = slot mod 16; same address or not makes little difference: scripts/ubench/lds_pattern_cost.hip), and a
    // bundle's footprint is a few nodes wide per axis: stored plainly, its nodes pile up on the banks of a few z values.
    //   * the 8 x 8 x 16 box (ROT): DENSE rows of 16 whose z index is SWIZZLED by x and y.  Scored with the measured
    //     cost law on oracle ray paths (scripts/deposit_layouts.py --accumulate [--xorsep]) both swizzles below equal the
    //     best padded layouts (16.4 / 16.6 cycles per add; rows of 18 / planes of 148: 16.9) in 8,192 B instead of
    //     9,472: with the 2 KB box B exactly the 10,240 B that give SIXTEEN waves per CU, the cap the registers set anyway.
    //       XORSW (the plain trace): slot = 128 x + 16 y + (z ^ (4 x & 15) ^ 2 y).  An exclusive-or whose x and y terms
    //         are separate makes a node's byte offset ONE three-input v_bitop3 of a per-x, a per-y and a per-z term
    //         (lds_add8): 22 address instructions per flush where the rotation takes 34 (12.70 against 12.95 ms).
    //       otherwise (the CBET kernels): slot = 128 x + 16 y + ((z + 7 x + 3 y) & 15), round 4's rotation.  They sit at
    //         128 registers; with the exclusive-or form the energy-field pass spills a constant of its phi series, reloads
    //         it inside the step and takes 22.2 ms instead of 20.2.
    //   * PAD without ROT (the 8 x 8 x 8 box of the field pass): rows of WZ + 1, planes padded by 4.
    //   * neither: dense, for the rarely used second box.
    static constexpr bool ROT = PAD && WZ == 16;
    static constexpr int YS = (PAD && !ROT) ? WZ + 1 : WZ;
    static constexpr int XS = (PAD && !ROT) ? WY * YS + 4 : WY * WZ;
    // the z index inside a row: tile coordinates in, position in the row out
    static constexpr bool XOR = ROT && XORSW;
    static __device__ __forceinline__ int zr(int tx, int ty, int tz)
    {
        return XOR ? (tz ^ ((tx & 3) << 2) ^ (ty << 1)) : (ROT ? ((tz + 7 * tx + 3 * ty) & ZM) : tz);
    }
    // ... as byte-offset terms whose exclusive-or is the slot's byte offset (x, y, z masked tile coordinates): the x term
    // carries x's field and (4 x & 15) << 3 -- one multiply, the stray third bit of x << 5 cleared --, the y term y's field
    // and 2 y << 3
    static_assert(!XOR || (WX == 8 && WY == 8 && WZ == 16), "the swizzle terms are written for the 8 x 8 x 16 box");
    static __device__ __forceinline__ int xterm(int tx) { return __mul24(tx, XS * 8 + 32) & ~0x80; }
    static __device__ __forceinline__ int yterm(int ty) { return __mul24(ty, YS * 8 + 16); }
    static constexpr int N = WX * XS;           // doubles per tile
    // largest offset of a lane's low corner from the origin at which its two nodes still lie inside
    static constexpr int SX = WX - 2, SY = WY - 2, SZ = WZ - 2;
    static constexpr bool BRICK = WZ == 16 && WY == 8;   // z follows in aligned bricks of 8 planes (rows of 8 x 8 lanes)
    static constexpr int DT = WX * WY * WZ;     // unpadded component tile (CBET field pass)
    static __device__ __forceinline__ int slot_d(int tx, int ty, int tz) { return (tx * WY + ty) * WZ + tz; }
};

// Wave-uniform state of one box: origin = haloed index of its low corner; it covers [o, o+W) per axis.
struct Origin {
    int x, y, z;
};

// absolute coordinate in [o, o + M + 1) whose residue modulo M + 1 is r's
template <int M>
__device__ __forceinline__ int abs_in(int o, int r) { return o + ((r - o) & M); }

struct WaveCounters {
    int n_atomics = 0;        // per lane
    int n_miss = 0;           // per lane: ray-steps deposited straight to HBM
    unsigned steps_miss = 0;  // wave-uniform, packed: wave-steps << 16 | wave-steps with a window miss
    unsigned slabs_bsteps = 0;// wave-uniform, packed: planes / bricks retired << 16 | wave-steps with box B live
    int pend = 0;             // wave-uniform: vector-memory instructions issued since the step's record gather
#ifdef CBET_DIAG_CLOCKS
    unsigned long long dg_ret = 0, dg_nret = 0;   // diagnostic build: shader clocks inside the write-backs of follow_box, their number
#endif
};

// The deposit grid as the write-back paths see it.  W32: the grid is smaller than 2^32 bytes (checked on the host,
// launch_trace_window), so a node's byte offset fits the 32-bit offset register of the scalar-base addressing mode -- one
// v_add_lshl_u32 in front of the atomic instead of an add, a sign extension and a 64-bit shift-add.
template <bool W32>
struct Grid {
    double *p;
};
template <bool W32>
__device__ __forceinline__ void grid_add(const TraceArgs &a, Grid<W32> g, int node, double v)
{
    if constexpr (W32) global_add(a, reinterpret_cast<double *>(reinterpret_cast<char *>(g.p) + ((unsigned)node * 8u)), v);
    else global_add(a, &g.p[node], v);
}
// ... at a 64-bit index (the components of the field pass lie whole per-beam arrays apart)
template <bool W32>
__device__ __forceinline__ void grid_add_far(const TraceArgs &a, Grid<W32> g, long index, double v) { global_add(a, &g.p[index], v); }

// Take the plane `coord` (absolute, inside the box) of axis AX (0: x, 1: y) out of a tile: read the sums, zero the
// non-zero ones and add them to HBM.  A plane is W x WZ entries (W = the other lateral extent), z fastest across
// lanes, so one wave instruction covers whole rows -- 64-B lines when the z origin is brick-aligned.  Every vector
// memory instruction that is really issued (some lane has a non-zero sum) is counted in wc.pend: the step's wait
// for its record gather skips exactly that many younger instructions (see the kernel).
template <class T, int AX, int NC, class G>
__device__ __forceinline__ void retire_plane(const TraceArgs &a, double *tile, const Origin &o, int coord, int lane,
                                             G edep, int sXh, int sYh, WaveCounters &wc, int coff, long gstride);

// COUNT neighbouring planes (coord, coord + step, ...) of a single-component tile at once: all reads first, one wait.
template <class T, int AX, int COUNT, class G>
__device__ __forceinline__ void retire_planes(const TraceArgs &a, double *tile, const Origin &o, int coord, int step,
                                              int lane, G edep, int sXh, int sYh, WaveCounters &wc)
{
    constexpr int WO = AX == 0 ? T::WY : T::WX;
    constexpr int IT = (WO * T::WZ + kWave - 1) / kWave;
    double v[COUNT * IT];
    int slot[COUNT * IT], node[COUNT * IT];
#pragma unroll
    for (int pl = 0; pl < COUNT; ++pl) {
        const int c = coord + pl * step;
        const int fixed = c & (AX == 0 ? T::XM : T::YM);
#pragma unroll
        for (int e = 0; e < IT; ++e) {
            const int idx = e * kWave + lane, r0 = idx / T::WZ, r1 = idx & T::ZM, k = abs_in<T::ZM>(o.z, r1);
            const int q = pl * IT + e;
            if (AX == 0) {
                slot[q] = fixed * T::XS + r0 * T::YS + T::zr(fixed, r0, r1);
                node[q] = c * sXh + abs_in<T::YM>(o.y, r0) * sYh + k;
            } else {
                slot[q] = r0 * T::XS + fixed * T::YS + T::zr(r0, fixed, r1);
                node[q] = abs_in<T::XM>(o.x, r0) * sXh + c * sYh + k;
            }
            const bool ok = (!(WO * T::WZ < kWave) || idx < WO * T::WZ) && CBET_AUDIT(a, (unsigned)slot[q] < (unsigned)T::N);
            v[q] = ok ? tile[slot[q]] : 0.0;
        }
    }
#pragma unroll
    for (int q = 0; q < COUNT * IT; ++q) {
        wc.pend += (CBET_BALLOT(v[q] != 0.0) != 0ull) ? 1 : 0;
        if (v[q] != 0.0) {
            tile[slot[q]] = 0.0;
            ++wc.n_atomics;
            grid_add(a, edep, node[q], v[q]);
        }
    }
}

template <class T, int AX, int NC, class G>
__device__ __forceinline__ void retire_plane(const TraceArgs &a, double *tile, const Origin &o, int coord, int lane,
                                             G edep, int sXh, int sYh, WaveCounters &wc, int coff, long gstride)
{
    constexpr int WO = AX == 0 ? T::WY : T::WX;      // extent of the other lateral axis
    constexpr int IT = (WO * T::WZ + kWave - 1) / kWave;
    const int fixed = coord & (AX == 0 ? T::XM : T::YM);
#pragma unroll
    for (int e = 0; e < IT; ++e) {
        const int idx = e * kWave + lane;
        const bool in_plane = !(WO * T::WZ < kWave) || idx < WO * T::WZ;
        const int r0 = idx / T::WZ, r1 = idx & T::ZM;
        int slot, node, slot_d;
        const int k = abs_in<T::ZM>(o.z, r1);
        if (AX == 0) {
            slot = fixed * T::XS + r0 * T::YS + T::zr(fixed, r0, r1);
            slot_d = T::slot_d(fixed, r0, r1);
            node = coord * sXh + abs_in<T::YM>(o.y, r0) * sYh + k;
        } else {
            slot = r0 * T::XS + fixed * T::YS + T::zr(r0, fixed, r1);
            slot_d = T::slot_d(r0, fixed, r1);
            node = abs_in<T::XM>(o.x, r0) * sXh + coord * sYh + k;
        }
        const bool ok = in_plane && CBET_AUDIT(a, (unsigned)slot < (unsigned)T::N);
        const double v = ok ? tile[slot] : 0.0;
        if (NC > 1) {
#pragma unroll
            for (int q = 1; q < NC; ++q) {
                const double vq = ok ? tile[coff + (q - 1) * T::DT + slot_d] : 0.0;
                wc.pend += (CBET_BALLOT(vq != 0.0) != 0ull) ? 1 : 0;
                if (vq != 0.0) {
                    grid_add_far(a, edep, q * gstride + node, vq);
                    tile[coff + (q - 1) * T::DT + slot_d] = 0.0;
                    ++wc.n_atomics;
                }
            }
        }
        wc.pend += (CBET_BALLOT(v != 0.0) != 0ull) ? 1 : 0;
        if (v != 0.0) {  // only nodes that received deposits are non-zero, hence valid
            tile[slot] = 0.0;
            ++wc.n_atomics;
            grid_add(a, edep, node, v);
        }
    }
}

// z, single planes (tiles without bricks): the plane is WX x WY (x, y) entries, at most one per lane, each in its
// own 64-B line of HBM.
template <class T, int NC, class G>
__device__ __forceinline__ void retire_zplane(const TraceArgs &a, double *tile, const Origin &o, int coord, int lane,
                                              G edep, int sXh, int sYh, WaveCounters &wc, int coff, long gstride)
{
    static_assert(T::WX * T::WY <= kWave, "one z-plane entry per lane");
    const int r0 = lane / T::WY, r1 = lane & T::YM, fixed = coord & T::ZM;
    const int slot = r0 * T::XS + r1 * T::YS + T::zr(r0, r1, fixed);
    const int node = abs_in<T::XM>(o.x, r0) * sXh + abs_in<T::YM>(o.y, r1) * sYh + coord;
    const bool ok = (!(T::WX * T::WY < kWave) || lane < T::WX * T::WY) && CBET_AUDIT(a, (unsigned)slot < (unsigned)T::N);
    const double v = ok ? tile[slot] : 0.0;
    if (NC > 1) {
        const int slot_d = T::slot_d(r0, r1, fixed);
#pragma unroll
        for (int q = 1; q < NC; ++q) {
            const double vq = ok ? tile[coff + (q - 1) * T::DT + slot_d] : 0.0;
            wc.pend += (CBET_BALLOT(vq != 0.0) != 0ull) ? 1 : 0;
            if (vq != 0.0) {
                grid_add_far(a, edep, q * gstride + node, vq);
                tile[coff + (q - 1) * T::DT + slot_d] = 0.0;
                ++wc.n_atomics;
            }
        }
    }
    wc.pend += (CBET_BALLOT(v != 0.0) != 0ull) ? 1 : 0;
    if (v != 0.0) {
        tile[slot] = 0.0;
        ++wc.n_atomics;
        grid_add(a, edep, node, v);
    }
}

// z, bricks (WZ = 16): the 8 planes [zb, zb + 8), zb a multiple of 8, leave together.  One wave instruction per
// tile x index; lanes = (y, z), z fastest: every atomic request is one full 64-B line of HBM.
template <class T, class G>
__device__ __forceinline__ void retire_zbrick(const TraceArgs &a, double *tile, const Origin &o, int zb, int lane,
                                              G edep, int sXh, int sYh, WaveCounters &wc)
{
    static_assert(T::WY == 8 && T::WZ == 16, "a brick is 8 rows of 8 planes per tile x index");
    const int ty = lane >> 3, kz = lane & 7;
    const int tz = (zb + kz) & T::ZM;
    const int base_node = abs_in<T::YM>(o.y, ty) * sYh + zb + kz;
    // all reads of a round first, one trip through the LDS queue per round (a read behind its own wait costs a trip each:
    // eight in a row were 2 % of the pass)
    constexpr int ROUND = 4;
#pragma unroll
    for (int t0 = 0; t0 < T::WX; t0 += ROUND) {
        double v[ROUND];
        int slot[ROUND];
#pragma unroll
        for (int q = 0; q < ROUND; ++q) {
            slot[q] = (t0 + q) * T::XS + ty * T::YS + T::zr(t0 + q, ty, tz);
            v[q] = CBET_AUDIT(a, (unsigned)slot[q] < (unsigned)T::N) ? tile[slot[q]] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < ROUND; ++q) {
            wc.pend += (CBET_BALLOT(v[q] != 0.0) != 0ull) ? 1 : 0;
            if (v[q] != 0.0) {
                tile[slot[q]] = 0.0;
                ++wc.n_atomics;
                grid_add(a, edep, abs_in<T::XM>(o.x, t0 + q) * sXh + base_node, v[q]);
            }
        }
    }
}

// Everything a box still holds goes to HBM (wave end, or box B emptying).
template <class T, int NC, class G>
__device__ __forceinline__ void flush_box(const TraceArgs &a, double *tile, const Origin &o, int lane, G edep,
                                          int sXh, int sYh, WaveCounters &wc, int coff, long gstride)
{
    for (int t = 0; t < T::WX; ++t)
        retire_plane<T, 0, NC>(a, tile, o, o.x + t, lane, edep, sXh, sYh, wc, coff, gstride);
}

// Wave-uniform decision of the hysteresis rule for one axis followed by planes: the number of planes to shift the
// origin by, negative = down, 0 = stay.  r = the lane's low-corner offset from the origin, mm = ballot of the box's
// member lanes, S = the largest offset at which a lane's two nodes still lie inside.  Shift when a member sits on
// an edge cell or outside, never shift a member out, never shift back on the next step.  Boxes at least 8 wide
// shift by TWO planes when every member would still sit a plane clear of the far edge afterwards: a bundle that
// advances along the axis then shifts every other cell (0.26 instead of 0.39 shifts per wave-step at 256^3), and a
// shift costs mostly its fixed part -- the decisions and a round trip through the LDS queue.
__device__ __forceinline__ int follow_plane_axis(int r, unsigned long long mm, int S)
{
    // The rule: with below / at_lo / near_lo = some member at r < 0 / <= 0 / <= 1 and above / at_hi / near_hi = some member
    // at r > S / >= S / >= S - 1, shift down iff (below or (at_lo and not near_hi)) and not at_hi, up iff (above or (at_hi
    // and not near_lo)) and not at_lo; by two planes when no member lies within two planes of the far edge.  Evaluated as
    // a decision tree -- four ballots on the usual path (a member on one edge, nobody near the other) instead of the
    // eight thresholds.
    const unsigned long long edge = __builtin_amdgcn_ballot_w64((unsigned)(r - 1) >= (unsigned)(S - 1)) & mm;   // r <= 0 or r >= S
    if (edge == 0ull) return 0;
    const unsigned long long hi = __builtin_amdgcn_ballot_w64(r >= S) & edge;     // at_hi (a member above the box included)
    const bool at_hi = hi != 0ull, at_lo = hi != edge;                            // (the edge lanes that are not high are low)
    if (at_hi == at_lo) return 0;                                                 // members on both edges: stay
    if (at_hi) {   // not at_lo, hence not below: up iff above or not near_lo
        if ((__builtin_amdgcn_ballot_w64(r <= 1) & mm) != 0ull) return (__builtin_amdgcn_ballot_w64(r > S) & mm) != 0ull ? 1 : 0;
        return (S >= 6 && (__builtin_amdgcn_ballot_w64(r <= 2) & mm) == 0ull) ? 2 : 1;   // (three planes at once, where the bundle is narrow enough, measured no better)
    }
    // at_lo only, hence not above: down iff below or not near_hi
    if ((__builtin_amdgcn_ballot_w64(r >= S - 1) & mm) != 0ull) return (__builtin_amdgcn_ballot_w64(r < 0) & mm) != 0ull ? -1 : 0;
    return (S >= 6 && (__builtin_amdgcn_ballot_w64(r >= S - 2) & mm) == 0ull) ? -2 : -1;
}

// ... and for z followed by aligned bricks (WZ = 16): the lane's two z nodes are r, r + 1 in [0, 16); shift by a
// brick when a member needs the next one and no member still needs the one that leaves.
__device__ __forceinline__ int follow_brick_axis(int r, unsigned long long mm)
{
    if ((__builtin_amdgcn_ballot_w64((unsigned)r > 14u) & mm) == 0ull) return 0;
    const bool below = (__builtin_amdgcn_ballot_w64(r < 0) & mm) != 0ull, above = (__builtin_amdgcn_ballot_w64(r > 14) & mm) != 0ull;
    const bool needs_lo = (__builtin_amdgcn_ballot_w64(r <= 7) & mm) != 0ull, needs_hi = (__builtin_amdgcn_ballot_w64(r >= 7) & mm) != 0ull;
    return (below && !needs_hi) ? -1 : ((above && !needs_lo) ? 1 : 0);
}

// Keep a box around its member lanes: (lx, ly, lz) = the lane's low corner (haloed), mm = ballot of the lanes
// that count for this box.  The decisions are taken first, as scalars; the planes that leave are then written
// back and the origin is moved by plain scalar arithmetic outside every divergent region (so that it stays in
// scalar registers).  Returns true when the origin moved.
template <class T, int NC, class G>
__device__ __forceinline__ bool follow_box(const TraceArgs &a, double *tile, Origin &o, unsigned long long mm, int lx,
                                           int ly, int lz, int lane, G edep, int sXh, int sYh, WaveCounters &wc,
                                           int coff, long gstride)
{
    auto leave = [&](auto axis, int d, int lo, int hi) {   // planes [lo, lo + |d|) or (hi - |d|, hi] leave along this axis
        constexpr int AX = decltype(axis)::value;
        const int first = d < 0 ? hi : lo, step = d < 0 ? -1 : 1, n = d < 0 ? -d : d;
        if constexpr (NC == 1) {
            if (n == 2) retire_planes<T, AX, 2>(a, tile, o, first, step, lane, edep, sXh, sYh, wc);
            else retire_planes<T, AX, 1>(a, tile, o, first, step, lane, edep, sXh, sYh, wc);   // (both reads of the plane first: one trip through the LDS queue)
        } else {
            for (int q = 0; q < n; ++q) retire_plane<T, AX, NC>(a, tile, o, first + q * step, lane, edep, sXh, sYh, wc, coff, gstride);
        }
    };
    const int dx = follow_plane_axis(lx - o.x, mm, T::SX);
    const int dy = follow_plane_axis(ly - o.y, mm, T::SY);
#ifdef CBET_DIAG_CLOCKS
    unsigned long long dg_a, dg_b;
    auto dg_in = [&]() { asm volatile("s_memtime %0" : "=&s"(dg_a) : : "memory"); };
    auto dg_out = [&]() {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dg_b) : "s"(dg_a) : "memory");
        wc.dg_ret += dg_b - dg_a;
        wc.dg_nret += 1;
    };
#else
    auto dg_in = []() {};
    auto dg_out = []() {};
#endif
    if (dx != 0) { dg_in(); leave(std::integral_constant<int, 0>{}, dx, o.x, o.x + T::WX - 1); dg_out(); }
    o.x += dx;
    if (dy != 0) { dg_in(); leave(std::integral_constant<int, 1>{}, dy, o.y, o.y + T::WY - 1); dg_out(); }
    o.y += dy;
    int dz;
    if constexpr (T::BRICK) {
        dz = 8 * follow_brick_axis(lz - o.z, mm);
        if (dz != 0) { dg_in(); retire_zbrick<T>(a, tile, o, dz < 0 ? o.z + 8 : o.z, lane, edep, sXh, sYh, wc); dg_out(); }
    } else {
        dz = follow_plane_axis(lz - o.z, mm, T::SZ);
        if (dz != 0)
            retire_zplane<T, NC>(a, tile, o, dz < 0 ? o.z + T::WZ - 1 : o.z, lane, edep, sXh, sYh, wc, coff, gstride);
        for (int q = 1; q < (dz < 0 ? -dz : dz); ++q)
            retire_zplane<T, NC>(a, tile, o, dz < 0 ? o.z + T::WZ - 1 - q : o.z + q, lane, edep, sXh, sYh, wc, coff, gstride);
    }
    o.z += dz;
    const bool moved = (dx | dy | dz) != 0;
    wc.slabs_bsteps += moved ? (1u << 16) : 0u;
    return moved;
}

template <class T>
__device__ __forceinline__ bool holds(const Origin &o, int lx, int ly, int lz)
{
    return (unsigned)(lx - o.x) <= (unsigned)T::SX && (unsigned)(ly - o.y) <= (unsigned)T::SY &&
           (unsigned)(lz - o.z) <= (unsigned)T::SZ;
}

// Is every cell a member of this box can occupy deep inside the grid (cbet_relocate.h, kRelocateDeep <= c <=
// n - 3) -- and therefore also more than two cells from every exit plane?  A held lane's low corner lies in
// [o, o + S], its cell index c is the low corner or one less.  Scalar arithmetic only.
template <class T>
__device__ __forceinline__ bool box_deep_inside(const Origin &o, int nx, int ny, int nz)
{
    // six differences that must all be >= 0: the OR of their sign bits in one compare (straight-line scalar code)
    const int lo = kRelocateDeep + 1;
    const int t = (o.x - lo) | (nx - 3 - T::SX - o.x) | (o.y - lo) | (ny - 3 - T::SY - o.y) | (o.z - lo) | (nz - 3 - T::SZ - o.z);
    return t >= 0;
}

// ---------------------------------------------------------------------------------------------
// The record gather and its wait, hand-scheduled.  Loads, stores and atomics share ONE in-order counter (vmcnt) on
// CDNA: a wait for a load also waits for every vector-memory instruction issued before it, and the compiler, which
// cannot count the conditionally issued write-back atomics of a step, waits with vmcnt(0) -- i.e. for the atomics
// issued AFTER the gather as well, a full round trip to the memory-side atomic unit in every step that moved a box.
// So the gather is issued from inline assembly (the compiler inserts no wait for it), the write-back paths count
// the vector-memory instructions they really issue (WaveCounters::pend, wave-uniform), and the step waits with the
// largest vmcnt(N), N <= pend, of a small ladder: the gather has arrived, the younger atomics stay in flight and have
// until the next step's wait -- a whole step -- to complete.
// ---------------------------------------------------------------------------------------------
typedef double dbl2 __attribute__((ext_vector_type(2)));

// WIDE = false: the table is at most 2^32 bytes (2^27 nodes, 512^3), so a record's byte offset fits the 32-bit
// offset register of the scalar-base addressing mode: one shift instead of 64-bit address arithmetic.
template <bool WIDE>
__device__ __forceinline__ void record_issue(const StepRecord *base, unsigned cell, dbl2 &kxy, dbl2 &kzk)
{
    // (the trailing comment names the destination registers in the assembly listing: tests/test_isa_audit.py checks
    // that the wait below names the same ones, i.e. that the compiler never moved the in-flight record)
    if constexpr (WIDE) {
        const char *p = reinterpret_cast<const char *>(base) + ((unsigned long long)cell << 5);
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\t; CBET_RECORD_ISSUE %0 %1"
                     : "=&v"(kxy), "=&v"(kzk)
                     : "v"(p)
                     : "memory");
    } else {
        const unsigned off = cell << 5;
        // (s_nop 4: the compiler may have reloaded `base` from a spill lane with v_readlane right in front of this block,
        // and a VALU write of an SGPR needs five wait states before a vector-memory instruction reads it -- a hazard
        // the compiler does not track into inline assembly: the audited build faulted on exactly that)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16\n\t; CBET_RECORD_ISSUE %0 %1"
                     : "=&v"(kxy), "=&v"(kzk)
                     : "v"(off), "s"(base)
                     : "memory");
    }
}

// pend = vector-memory instructions issued since record_issue (exact, or an underestimate -- never more).  The common
// case -- nothing younger in flight: 75 % of the wave-steps -- is one compare and one taken branch.
__device__ __forceinline__ void record_wait(dbl2 &kxy, dbl2 &kzk, int pend)
{
    asm volatile("; CBET_RECORD_WAIT %0 %1\n\t"
                 "s_cmp_eq_u32 %2, 0\n\t"
                 "s_cbranch_scc1 .Lrw_0_%=\n\t"
                 "s_cmp_ge_u32 %2, 8\n\t"
                 "s_cbranch_scc1 .Lrw_8_%=\n\t"
                 "s_cmp_ge_u32 %2, 4\n\t"
                 "s_cbranch_scc1 .Lrw_4_%=\n\t"
                 "s_cmp_ge_u32 %2, 2\n\t"
                 "s_cbranch_scc1 .Lrw_2_%=\n\t"
                 "s_waitcnt vmcnt(1)\n\t"
                 "s_branch .Lrw_end_%=\n"
                 ".Lrw_2_%=:\n\t"
                 "s_waitcnt vmcnt(2)\n\t"
                 "s_branch .Lrw_end_%=\n"
                 ".Lrw_4_%=:\n\t"
                 "s_waitcnt vmcnt(4)\n\t"
                 "s_branch .Lrw_end_%=\n"
                 ".Lrw_8_%=:\n\t"
                 "s_waitcnt vmcnt(8)\n\t"
                 "s_branch .Lrw_end_%=\n"
                 ".Lrw_0_%=:\n\t"
                 "s_waitcnt vmcnt(0)\n"
                 ".Lrw_end_%=:"
                 : "+v"(kxy), "+v"(kzk)
                 : "s"(pend)
                 : "memory", "scc");
}

// ... and the plain form: wait for every vector-memory instruction in flight (steps that issued nothing behind the gather)
__device__ __forceinline__ void record_wait_all(dbl2 &kxy, dbl2 &kzk)
{
    asm volatile("; CBET_RECORD_WAIT %0 %1\n\ts_waitcnt vmcnt(0)" : "+v"(kxy), "+v"(kzk) : : "memory");
}

// a + b + c in one instruction (the compiler, left alone, shares partial sums instead: more instructions)
__device__ __forceinline__ int add3(int a, int b, int c)
{
    int r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// (f ^ 1) + c in one instruction: the node a lane visits first along an axis, own + 1 - flip (keeping 1 - flip in a register
// of its own costs one, forming it every step an instruction)
__device__ __forceinline__ int xad1(int f, int c)
{
    int r;
    asm("v_xad_u32 %0, %1, 1, %2" : "=v"(r) : "v"(f), "v"(c));
    return r;
}

// max(|a|, |b|, |c|) in two instructions (fmax() adds a canonicalising v_max_f64 x, x per operand); NaNs are ignored
__device__ __forceinline__ double max_abs3(double a, double b, double c)
{
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(r), "v"(c));
    return r;
}

// var = value through a move TIED to var's register: the new value lives where the old one did (see the window arm)
__device__ __forceinline__ void commit(int &var, int value)
{
    asm volatile("s_mov_b32 %0, %1" : "+s"(var) : "s"(__builtin_amdgcn_readfirstlane(value)));   // (wave-uniform by construction)
}
__device__ __forceinline__ void commit(unsigned long long &var, unsigned long long value)
{
    asm volatile("s_mov_b64 %0, %1" : "+s"(var) : "s"(value));
}
__device__ __forceinline__ void commit_v(int &var, int value) { asm volatile("v_mov_b32 %0, %1" : "+v"(var) : "v"(value)); }

// a * b + c on 24-bit operands (the compiler picked the quarter-rate v_mad_u64_u32 for one of the two).  b is a scalar
// register: on gfx940 / gfx950 a vector instruction that reads an SGPR needs two wait states after a vector instruction
// that wrote it (a v_readlane reloading it from a spill lane), a hazard the compiler tracks for its own instructions
// but not into inline assembly -- hence the s_nop.
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int r;
    asm("s_nop 1\n\tv_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// ---------------------------------------------------------------------------------------------
// The kernel.
// ---------------------------------------------------------------------------------------------

#ifdef CBET_DEBUG_BOUNDS
constexpr bool kAudited = true;
#else
constexpr bool kAudited = false;
#endif
#ifdef CBET_DIAG_CLOCKS
constexpr bool kDiagClocks = true;
#else
constexpr bool kDiagClocks = false;
#endif
constexpr double kNearTol = 0.5001;   // launch_ray_XZ.cu:132, the nearest-node tolerance
constexpr double kFarJump = 1.4998;   // relocate_deep_interior's validity bound on |f - cell|

// a per-lane predicate FROM a wave-uniform 64-bit mask (no instruction: the mask is used as the condition register)
#define CBET_LANES(mask) __builtin_amdgcn_inverse_ballot_w64(mask)

// Lane predicates of the step loop are kept as 64-bit masks in scalar registers and combined there explicitly: `live`
// (the ray is still traced), `hbm` (its home is box B), `miss_m` (its pending deposit goes straight to HBM instead of an
// LDS box).  A per-lane copy of a predicate costs the compiler merges under the exec mask wherever it
// crosses a branch (three scalar instructions each); a mask costs nothing until it is used (CBET_LANES).
//
// STATS: the window diagnostics (wave-steps, misses, box-B steps, planes retired, global atomics) are counted only by
// this instantiation (cbet_params.window_stats); the other counts ray-steps and rays.
template <int WZ, bool GENERIC, int CBET, bool STATS>
__global__ void __launch_bounds__(kWave, (CBET == 4) ? 1 : 4) k_trace_window(const TraceArgs a)
{
    using T = Tile<8, 8, WZ, true, CBET == 0>;   // box A
    // Box B holds the few lanes that left A: 4 x 8 x 8 nodes in the dense layout, 2 KB.  Occupancy is what this
    // loop responds to (256^3 pass, round 2: 25.7 ms at 8 waves per CU, 21.9 at 11, 20.9 at 12, 19.3 at 14; round 4: 16.3
    // at 14, 15.7 at 16), and the LDS is what caps it: a 4 KB B (8 x 8 x 8) halves the window misses but costs more than
    // it saves, and A stored plainly costs 35-40 LDS cycles per ds_add_f64 (27 ms in round 2).  Shapes of the 2 KB, 256^3
    // pass (round 2): 8x8x4 19.26 ms (misses 1.00 % of ray-steps), 8x4x8 19.09 (0.87 %), 4x8x8 18.81 (0.86 %), 4x4x16
    // 19.11 (1.04 %), 4x16x4 / 16x4x4 19.7 (1.18 %); placing a new B off-centre towards A changes nothing.
    using TB = Tile<4, 8, 8, false>;
    constexpr bool IDX64 = GENERIC;
    constexpr int NC = (CBET == 4) ? 4 : 1;
    constexpr bool ACC = (CBET == 0);   // deposits summed in registers until the ray's nodes change (see `accumulate`)
    // ... and the addition itself runs in the NEXT step's gather shadow (the factors and the increment cross the loop edge):
    // 23 vector instructions off the chain wait -> absorb -> move -> relocate -> gather (13.09 against 13.17 ms)
    constexpr bool PIPE = ACC;
    constexpr int NSLOT = T::N + TB::N;                   // box A, box B
    constexpr int NLDS = NSLOT + (NC - 1) * T::DT;        // + components 1.. of box A (field pass)
    __shared__ double s_val[NLDS];
    const int lane = threadIdx.x;

    int beam, patch;
    if (!work_item(a, blockIdx.x, beam, patch)) return;
    // beam-resolved deposition (cbet_params.per_beam_grids): beam b accumulates into its own grid,
    // edep[b * grid_stride ...]; otherwise every beam adds into the one grid (grid_stride = 0)
    const Grid<!IDX64> edep{a.edep + (long)(beam - a.grid_beam0) * a.grid_stride};
    const bool absorb = GENERIC ? (a.absorption == 1) : true;   // def.cuh:118

    Ray s = {};   // holes and culled rays keep zeros: their lanes run the arithmetic below on harmless values
    const int li = patch * kWave + lane;
    const int pre_raynum = li < a.nlive ? a.live[li] : -1;  // -1: hole in the 8x8 patch
    bool launched = pre_raynum >= 0;
    if (launched) launched = launch_ray(a, beam, pre_raynum, s);
    unsigned long long live = __ballot(launched);
    if (live == 0ull) return;  // whole bundle culled (cannot happen for a listed patch; cheap guard)
    const int tot_rays = __popcll(live);

    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const int sYh = a.sYh, sXh = a.sXh;                   // haloed edep strides (:5-7; rows of nz+2 doubles unless the caller's grid is padded)
    unsigned cell = launched ? (unsigned)((s.ci * ny + s.cj) * nz + s.ck) : 0u;
    double fcx = (double)s.ci, fcy = (double)s.cj, fcz = (double)s.ck;   // the cell as the reference's (double)thisx
    int tot_steps = 0;              // wave-uniform: ray-steps of this bundle (a lane's count is added when it ends)
    WaveCounters wc;

    double *const tileA = s_val, *const tileB = s_val + T::N;
    Origin oA{0, 0, 0}, oB{0, 0, 0};
    unsigned long long hbm = 0ull; // lanes whose deposits go to box B
    int b_active = 0;              // wave-uniform flag: box B holds lanes
    {
        const int src = ((live >> 27) & 1ull) ? 27 : (__ffsll((long long)live) - 1);
        for (int z = lane; z < NLDS; z += kWave) s_val[z] = 0.0;
        oA.x = __builtin_amdgcn_readlane(s.ci, src) + 1 - T::WX / 2;
        oA.y = __builtin_amdgcn_readlane(s.cj, src) + 1 - T::WY / 2;
        oA.z = __builtin_amdgcn_readlane(s.ck, src) + 1 - 4;
        if (T::BRICK) oA.z &= ~7;
        __syncthreads();
    }
    // wave-uniform flag: every live lane was held by a box after the last step and both boxes lie deep inside the grid
    int deep = 0;
#ifdef CBET_DIAG_CLOCKS
    // diagnostic build (never shipped): shader-clock cycles this wave spends in the record wait and in the window-shift
    // path, reported through the counter slots (see the end of the kernel)
    unsigned long long dg_wait = 0, dg_shift = 0, dg_nshift = 0, dg_t_start;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dg_t_start) : : "memory");
#endif

    // A step's record (cbet_device.h StepRecord: the three kicks and the absorption coefficient at the ray's node) is
    // gathered as soon as the new node is known; the absorption coefficient is used at the end of the same step,
    // the kicks by the NEXT step's move.  One aligned 32-byte gather per lane and step.
    dbl2 rec_kxy, rec_kzk;                   // {kx, ky}, {kz, kappa} of the ray's node
    auto gather_record = [&]() {             // all lanes (a dead lane reads node 0); see record_issue
        unsigned c = CBET_LANES(live) ? cell : 0u;
        asm("" : "+v"(c));   // (keeps the select a 32-bit one, in front of the address arithmetic)
#ifdef CBET_DEBUG_BOUNDS
        if (!(c < a.audit_nodes)) { audit_fail(a); c = 0u; }
#endif
        record_issue<IDX64>(a.steprec, c, rec_kxy, rec_kzk);
        wc.pend = 0;
    };
    // The wait, counted: the vector-memory instructions the step really issued behind the gather (the window pass's
    // write-backs, a flush's atomics) stay in flight.  CBET hooks and audited builds issue vector loads the compiler
    // tracks itself: they wait for everything.  ONE wait site in the loop, behind the join of the window logic: with a
    // wait in each arm the register allocator is free to give the record different registers in the arms and to copy
    // it -- before it has arrived -- where they meet (tests/test_isa_audit.py).
    auto await_record = [&]() {
#ifdef CBET_DIAG_CLOCKS
        unsigned long long t0, t1;
        asm volatile("s_memtime %0" : "=&s"(t0) : : "memory");
#endif
        if (CBET == 0 && !kAudited) record_wait(rec_kxy, rec_kzk, wc.pend);
        else record_wait_all(rec_kxy, rec_kzk);
#ifdef CBET_DIAG_CLOCKS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t1) : "s"(t0) : "memory");
        dg_wait += t1 - t0;
#endif
    };
    gather_record();
    await_record();
    const double *const gk = CBET && a.gain ? a.gain + (long)(beam - a.grid_beam0) * a.hsize : nullptr;  // this beam's gain grid
    double gained = 0.0;                     // CBET: energy this lane's ray gained

    // lane-dependent corner order (see the weights): which of an axis's two nodes a lane visits first
    const bool flx = (lane & 1) != 0, fly = (lane & 2) != 0, flz = (lane & 8) != 0;
    const int pfx = flx ? 1 : 0, pfy = fly ? 1 : 0, pfz = flz ? 1 : 0;

    // The per-lane arithmetic of a step runs on ALL lanes, dead ones included (their state is garbage nobody reads):
    // only memory accesses and the LDS / HBM adds are predicated.  Guarding the arithmetic with `if (alive)` costs
    // exec-mask bookkeeping plus, for every value that crosses the window logic, a merge of "this step's" and "the
    // dead lane's last" copy -- ~10 moves per step.
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;   // CBET = 4: the four field quantities a step deposits

    bool slow = true;                        // wave-uniform: this step runs the general (face-aware) forms
    // The loop is rotated: the dependent chain of a step is wait -> kick -> move -> relocate -> gather, everything else
    // fills the gather's shadow.
    double Fx0 = 0, Fx1 = 0, Fy0 = 0, Fy1 = 0, Fz0 = 0, Fz1 = 0;   // the step's six per-axis factors
    int X0 = 0, X1 = 0, Y0 = 0, Y1 = 0, Z0 = 0, Z1 = 0;            // the nodes the lane's pending sums belong to (haloed)
    double inc = 0.0;                        // :305-311 the energy the step deposits
    // Where a live lane's pending deposit belongs: an LDS box -- the tile at tile_off (in doubles; per lane, 0 while box B
    // is idle) -- unless the lane is in miss_m (outside both boxes after the last window pass): then straight to HBM.
    // (A ray that ends hands its deposit over at once, so only live lanes ever hold one.)
    int tile_off = 0;
    unsigned long long miss_m = 0ull;
    int own_slot = 0, own_node = 0;          // CBET = 4: the ray's own node, in box A's component tiles / in the grid
    // Eight sums to the lane's eight nodes X0..Z1 in LDS.  slot = (x & XM) * XS + (y & YM) * YS + zr with the masks and
    // strides of the lane's tile (byte offsets throughout); zr = the z index, swizzled by x and y in box A (Tile::zr).
    auto lds_add8 = [&](const double *w) {
        auto add = [&](int byte, double v) {
            if (CBET_AUDIT(a, (unsigned)byte < (unsigned)NSLOT * 8u))
                __hip_atomic_fetch_add(static_cast<double *>(__builtin_assume_aligned(reinterpret_cast<char *>(s_val) + byte, 8)),
                                       v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if constexpr (T::XOR) {
            // Both boxes are dense with power-of-two strides here: x, y and the z index are disjoint bit fields of the byte
            // offset, and box A's swizzle is an exclusive-or of an x and a y term into the z field -- a node's offset is the
            // exclusive-or of a per-x, a per-y and a per-z term: one v_bitop3_b32 (truth table 0x96) per node.
            int ax0, ax1, by0, by1, cz0, cz1;
            if (b_active == 0) {   // scalar branch: everything goes to box A, compile-time masks and strides
                ax0 = T::xterm(X0 & T::XM); ax1 = T::xterm(X1 & T::XM);
                by0 = T::yterm(Y0 & T::YM); by1 = T::yterm(Y1 & T::YM);
                cz0 = (Z0 & T::ZM) << 3; cz1 = (Z1 & T::ZM) << 3;
            } else {               // a lane of box B: the plain dense 4 x 8 x 8 tile behind box A (tile_off, in doubles)
                const bool toB = tile_off != 0;
                const int xm = toB ? TB::XM : T::XM, zm = toB ? TB::ZM : T::ZM;
                const int xmul = toB ? TB::XS * 8 : T::XS * 8 + 32, ymul = toB ? TB::YS * 8 : T::YS * 8 + 16;
                const int keep = toB ? ~0 : ~0x80, off8 = tile_off * 8;
                static_assert(TB::YM == T::YM, "one y mask for both boxes");
                ax0 = (__mul24(X0 & xm, xmul) & keep) | off8; ax1 = (__mul24(X1 & xm, xmul) & keep) | off8;
                by0 = __mul24(Y0 & T::YM, ymul); by1 = __mul24(Y1 & T::YM, ymul);
                cz0 = (Z0 & zm) << 3; cz1 = (Z1 & zm) << 3;
            }
            auto at = [](int ax, int by, int cz) { return (int)__builtin_amdgcn_bitop3_b32(ax, by, cz, 0x96); };
            add(at(ax0, by0, cz0), w[0]);
            add(at(ax1, by0, cz0), w[1]);
            add(at(ax0, by0, cz1), w[2]);
            add(at(ax1, by0, cz1), w[3]);
            add(at(ax0, by1, cz0), w[4]);
            add(at(ax1, by1, cz0), w[5]);
            add(at(ax0, by1, cz1), w[6]);
            add(at(ax1, by1, cz1), w[7]);
        } else {
            // 24-bit multiplies by the byte strides; rot = 1: the z index rotated by 7 x + 3 y (box A of the CBET kernels)
            auto add8 = [&](int xm, int ym, int zm, int xs, int ys, int off, int rot) {
                const int xa = X0 & xm, xb = X1 & xm, ya = Y0 & ym, yb = Y1 & ym;
                const int x0 = __mul24(xa, xs * 8) + off * 8, x1 = __mul24(xb, xs * 8) + off * 8;
                const int y0 = __mul24(ya, ys * 8), y1 = __mul24(yb, ys * 8);
                const int ra = 7 * xa * rot, rb = 7 * xb * rot, sa = 3 * ya * rot, sb = 3 * yb * rot;
                auto zb = [&](int z, int r) { return ((z + r) & zm) * 8; };
                if constexpr (T::ROT) {
                    // both boxes are dense with power-of-two strides here: x, y and the z index are disjoint bit fields of the
                    // byte offset, so a node's offset is (z term & mask) | (x | y) -- one v_and_or_b32 behind the z sum
                    const int xy00 = x0 | y0, xy10 = x1 | y0, xy01 = x0 | y1, xy11 = x1 | y1, zm8 = zm * 8;
                    auto at = [&](int xy, int z, int r) { return (((z + r) * 8) & zm8) | xy; };
                    add(at(xy00, Z0, ra + sa), w[0]);
                    add(at(xy10, Z0, rb + sa), w[1]);
                    add(at(xy00, Z1, ra + sa), w[2]);
                    add(at(xy10, Z1, rb + sa), w[3]);
                    add(at(xy01, Z0, ra + sb), w[4]);
                    add(at(xy11, Z0, rb + sb), w[5]);
                    add(at(xy01, Z1, ra + sb), w[6]);
                    add(at(xy11, Z1, rb + sb), w[7]);
                } else {
                    add(add3(x0, y0, zb(Z0, ra + sa)), w[0]);
                    add(add3(x1, y0, zb(Z0, rb + sa)), w[1]);
                    add(add3(x0, y0, zb(Z1, ra + sa)), w[2]);
                    add(add3(x1, y0, zb(Z1, rb + sa)), w[3]);
                    add(add3(x0, y1, zb(Z0, ra + sb)), w[4]);
                    add(add3(x1, y1, zb(Z0, rb + sb)), w[5]);
                    add(add3(x0, y1, zb(Z1, ra + sb)), w[6]);
                    add(add3(x1, y1, zb(Z1, rb + sb)), w[7]);
                }
            };
            if (b_active == 0) {   // scalar branch: everything goes to box A, compile-time masks and strides
                add8(T::XM, T::YM, T::ZM, T::XS, T::YS, 0, T::ROT ? 1 : 0);
            } else {
                const bool toB = tile_off != 0;
                add8(toB ? TB::XM : T::XM, toB ? TB::YM : T::YM, toB ? TB::ZM : T::ZM, toB ? TB::XS : T::XS,
                     toB ? TB::YS : T::YS, tile_off, (T::ROT && !toB) ? 1 : 0);
            }
        }
    };
    // eight values to the lane's eight nodes X0..Z1 in HBM (a lane outside both boxes)
    auto hbm_add8 = [&](const double *w) {
        const int nX0 = __mul24(X0, sXh), nX1 = __mul24(X1, sXh), nY0 = __mul24(Y0, sYh), nY1 = __mul24(Y1, sYh);
        grid_add(a, edep, nX0 + nY0 + Z0, w[0]);
        grid_add(a, edep, nX1 + nY0 + Z0, w[1]);
        grid_add(a, edep, nX0 + nY0 + Z1, w[2]);
        grid_add(a, edep, nX1 + nY0 + Z1, w[3]);
        grid_add(a, edep, nX0 + nY1 + Z0, w[4]);
        grid_add(a, edep, nX1 + nY1 + Z0, w[5]);
        grid_add(a, edep, nX0 + nY1 + Z1, w[6]);
        grid_add(a, edep, nX1 + nY1 + Z1, w[7]);
        wc.n_atomics += 8;
    };

    // ---- deposits, the CBET kernels (ACC = false): every step's deposit goes to LDS during the NEXT step, in the shadow of
    // that step's record gather (:341-348: a_c * increment to the eight nodes, a_c = (Fz * Fy) * Fx).  What it needs
    // crosses the loop edge: the six per-axis factors, the node indices, the increment, where it goes.  (Their gain
    // hooks leave no registers for the pending sums below: with them the energy-field pass spills and takes 33 ms
    // instead of 25.)
    auto deposit_step = [&](unsigned long long lds_m, unsigned long long hbm_m) {
        // 14 products instead of the reference's 20: ((Fz * inc) * Fy) * Fx for ((Fz * Fy) * Fx) * inc -- three roundings
        // either way, i.e. a deposit differs from the reference's by at most 2 ulp (the sum order of the atomics already
        // moves a cell's total by more: SURVEY 8(c)'s metric is 1e-9).  The ray's own state (position, velocity, energy,
        // cell: everything that decides where it goes and when it stops) keeps the reference's operations one for one.
        const double zi0 = Fz0 * inc, zi1 = Fz1 * inc;
        const double zy00 = zi0 * Fy0, zy10 = zi1 * Fy0, zy01 = zi0 * Fy1, zy11 = zi1 * Fy1;
        // order (x,y,z) = (0,0,0) (1,0,0) (0,0,1) (1,0,1) (0,1,0) (1,1,0) (0,1,1) (1,1,1) -- :341-348 without the flips
        double wgt[8];
        wgt[0] = zy00 * Fx0;
        wgt[1] = zy00 * Fx1;
        wgt[2] = zy10 * Fx0;
        wgt[3] = zy10 * Fx1;
        wgt[4] = zy01 * Fx0;
        wgt[5] = zy01 * Fx1;
        wgt[6] = zy11 * Fx0;
        wgt[7] = zy11 * Fx1;
        if (CBET_LANES(lds_m)) lds_add8(wgt);
        // window misses: eight atomics, younger than the record gather just issued -- counted for its wait
        if (hbm_m != 0ull) {
            wc.pend += 8;
            if (CBET_LANES(hbm_m)) {
                hbm_add8(wgt);
                ++wc.n_miss;
            }
        }
        if (CBET == 4) {
            // Displacement components: the ray's own node only -- box A's tiles, or HBM for a lane of
            // box B / outside the boxes.
            const bool inbox = CBET_LANES(lds_m);
            if (inbox && tile_off == 0) {
                if (CBET_AUDIT(a, (unsigned)(own_slot + 2 * T::DT) < (unsigned)NLDS)) {
                    __hip_atomic_fetch_add(&s_val[own_slot], q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_val[own_slot + T::DT], q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&s_val[own_slot + 2 * T::DT], q3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else if (inbox || CBET_LANES(hbm_m)) {
                grid_add_far(a, edep, a.comp_stride + own_node, q1);
                grid_add_far(a, edep, 2 * a.comp_stride + own_node, q2);
                grid_add_far(a, edep, 3 * a.comp_stride + own_node, q3);
                wc.n_atomics += 3;
            }
        }
    };

    // ---- deposits, the plain trace (ACC = true): summed in registers while the ray's eight nodes stay the same ----------
    // The deposit of a step (:341-348: a_c * increment to the eight nodes, a_c = (Fz * Fy) * Fx) is ADDED TO THE LANE'S
    // PENDING SUMS S[8] at the end of the step; the sums go to LDS (or, for a lane outside both boxes, to HBM) only when
    // the ray's low corner changes, or the ray ends.  A ray keeps its eight nodes for 1.9 steps on average (256^3), so
    // a ds_add_f64 carries 47 % of the lanes instead of all of them: the LDS pipeline serves 133 instead of 188 cycles per
    // wave-step (scripts/deposit_layouts.py).  The flush happens in the NEXT step, right after the new cell is known (in
    // the shadow of its record gather) and BEFORE that step's window pass moves anything, i.e. while the boxes still
    // stand where the last window pass put them for exactly these nodes.
    double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // pending sums for the nodes X0..Z1
    unsigned long long restart_m = 0ull;     // lanes whose sums went out in this step's flush and restart with its deposit
    bool p_odd = true;                       // wave-uniform: the nodes came from the rare (non-negative offset) branch
    auto accumulate = [&]() {
        // 14 products instead of the reference's 20 (see deposit_step).  S * 1 + w and S * 0 + w are exact forms of
        // "S + w" and "w": one fma each, written IN PLACE (left to the compiler it becomes v_fmac into the product's
        // register and eight 64-bit copies back at the loop edge)
        const double zi0 = Fz0 * inc, zi1 = Fz1 * inc;
        const double zy00 = zi0 * Fy0, zy10 = zi1 * Fy0, zy01 = zi0 * Fy1, zy11 = zi1 * Fy1;
        const double k = CBET_LANES(restart_m) ? 0.0 : 1.0;
        auto add = [&](double &sum, double w) { asm("v_fma_f64 %0, %0, %1, %2" : "+v"(sum) : "v"(k), "v"(w)); };
        add(S[0], zy00 * Fx0);
        add(S[1], zy00 * Fx1);
        add(S[2], zy10 * Fx0);
        add(S[3], zy10 * Fx1);
        add(S[4], zy01 * Fx0);
        add(S[5], zy01 * Fx1);
        add(S[6], zy11 * Fx0);
        add(S[7], zy11 * Fx1);
    };
    // the pending sums of the lanes in lds_m go to their LDS box, those of the lanes in hbm_m straight to HBM
    auto flush_sums = [&](unsigned long long lds_m, unsigned long long hbm_m) {
        if (CBET_LANES(lds_m)) lds_add8(S);
        // window misses: eight atomics, younger than the record gather just issued -- counted for its wait
        if (hbm_m != 0ull) {
            wc.pend += 8;
            if (CBET_LANES(hbm_m)) hbm_add8(S);
        }
    };

    // ---- the rest of a step: wait for the record, absorb, add the deposit to the pending sums, end rays -------------
    auto step_tail = [&](int &tt) {
    // ---- absorption (:305-311) ---------------------------------------------------------------------
    await_record();   // the record gathered after the relocation: kappa now, the kicks at the top of the next step
    if (absorb) {
        inc = rec_kzk.y * s.uray;
        s.uray -= inc;
    } else {
        inc = s.uray;
    }
    if (CBET >= 2) inc = q0;   // field passes deposit energy x path length
    if constexpr (ACC && !PIPE) accumulate();   // the step's deposit joins the lane's pending sums
    // ---- termination (:351-356) --------------------------------------------------------------------
    // The six exit planes are compared only when the wave is not deep inside the grid: a lane held by a deep
    // box is more than two cells from every face, far beyond the half cell of :352-354.  Ballots of plain
    // compares over all lanes, masked with `live` on the scalar unit.
    unsigned long long died = CBET_BALLOT(s.uray <= s.ustop);
    if (slow || deep == 0) {   // scalar branch
        const double *b = a.bounds;  // {xlo, xhi, ylo, yhi, zlo, zhi}
        died |= CBET_BALLOT(s.px < b[0]) | CBET_BALLOT(s.px > b[1]) | CBET_BALLOT(s.py < b[2]) |
                CBET_BALLOT(s.py > b[3]) | CBET_BALLOT(s.pz < b[4]) | CBET_BALLOT(s.pz > b[5]);
    }
    died &= live;
    if (died != 0ull) {   // scalar branch: some ray ended in this step, its tt + 1-th
        asm volatile("");
        commit(tot_steps, tot_steps + __popcll(died) * (tt + 1));
        // its deposit goes where the window pass of this step put it, now (the boxes stand for exactly these nodes)
        if constexpr (PIPE) {
            // (the step's deposit has not joined the sums yet: it does now, for every lane, and the next step's addition
            // becomes S * 1 + 0 for the lanes that go on)
            accumulate();
            asm volatile("v_mov_b64 %0, 0" : "+v"(inc));
            commit(restart_m, 0ull);
        }
        if constexpr (ACC) flush_sums(died & ~miss_m, died & miss_m);
        else deposit_step(died & ~miss_m, died & miss_m);
        commit(live, live & ~died);
        commit(hbm, hbm & live);
        commit(miss_m, miss_m & live);
        // (every ray has ended: the loop ends through its one exit, the step count -- a second exit costs the common path
        // the copies and flags of the compiler's exit unification)
        if (live == 0ull) commit(tt, a.nt - 1);
    }
    __builtin_amdgcn_wave_barrier();
    };

    for (int tt = 0; tt < a.nt; ++tt) {                        // :207  (live != 0 here: checked where lanes end)
        if constexpr (STATS || kDiagClocks) wc.steps_miss += 1u << 16;
        // ---- move ------------------------------------------------------------------------------------------
        // :268-273 kick then drift (stencil values gathered during the previous step)
        s.vx -= rec_kxy.x;
        s.vy -= rec_kxy.y;
        s.vz -= rec_kzk.x;
        s.px += s.vx * a.dt;
        s.py += s.vy * a.dt;
        s.pz += s.vz * a.dt;
        // :276-278 position in cell units
        const double fx = (s.px - a.xmin) * a.inv_dx;
        const double fy = (s.py - a.ymin) * a.inv_dy;
        const double fz = (s.pz - a.zmin) * a.inv_dz;
        // :282-292 nearest-node update, deep-interior form (cbet_relocate.h relocate_deep_interior: exact for
        // kRelocateDeep <= cell <= n-3 unless the ray moved more than a cell, which sends the wave to the closed form)
        const double g0x = fx - fcx, g0y = fy - fcy, g0z = fz - fcz;
        const bool upx = g0x >= kNearTol, dnx = g0x < kNearTol - 1.0, upy = g0y >= kNearTol, dny = g0y < kNearTol - 1.0,
                   upz = g0z >= kNearTol, dnz = g0z < kNearTol - 1.0;
        // wave-uniform: this step runs the general (face-aware) forms.  One compare of the largest |g| (a NaN -- which
        // moves no cell in either form -- is ignored by the maximum)
        slow = deep == 0 || (CBET_BALLOT(!(max_abs3(g0x, g0y, g0z) < kFarJump)) & live) != 0ull;
        // ---- relocate, gather -------------------------------------------------------------------
        // The deep-interior form updates the cell IN PLACE; the rare general form takes the update back first (kept as
        // copies for it, the old cell costs the common path three moves).
        const int di = (upx ? 1 : 0) - (dnx ? 1 : 0), dj = (upy ? 1 : 0) - (dny ? 1 : 0), dk = (upz ? 1 : 0) - (dnz ? 1 : 0);
        s.ci += di;
        s.cj += dj;
        s.ck += dk;
        if (slow) {                        // near a face (or a far jump): closed form with the candidate bounds
            int oi = s.ci - di, oj = s.cj - dj, ok = s.ck - dk;
            asm volatile("" : "+v"(oi), "+v"(oj), "+v"(ok));   // (recomputed here, not carried from above the update)
            s.ci = relocate_closed(oi, fx, nx);
            s.cj = relocate_closed(oj, fy, ny);
            s.ck = relocate_closed(ok, fz, nz);
        }
        const unsigned new_cell = (unsigned)mad24(mad24(s.ci, ny, s.cj), nz, s.ck);
        // lanes whose ray changed cell (ACC: their pending sums must leave): one compare with the cell it had
        unsigned long long moved_m = 0ull;
        if constexpr (ACC) moved_m = CBET_BALLOT(new_cell != cell);
        cell = new_cell;
        // :296-298 absorption coefficient at the new node and the NEXT step's kicks
        gather_record();
        // (everything below reads the cell through this barrier, i.e. is scheduled BEHIND the gather's issue: left alone the
        // compiler puts the offsets' nine instructions in front of it, on the dependent chain)
        asm volatile("" : "+v"(s.ci), "+v"(s.cj), "+v"(s.ck));
        fcx = (double)s.ci;
        fcy = (double)s.cj;
        fcz = (double)s.ck;
        if constexpr (!ACC) deposit_step(live & ~miss_m, miss_m);   // the previous step's deposit, in the shadow of the gather
        if constexpr (PIPE) accumulate();                           // ... or its addition to the pending sums
        // ---- weights (:319-339) -----------------------------------------------------------------
        // Each weight is (Fz * Fy) * Fx * inc with F = (1-d) for the ray's own node along that axis and F = d
        // for the neighbour on the `sign` side (:329-336).  The neighbour lies on the side of the offset's sign
        // (:338-339), so a lane's two nodes per axis are {low, low + 1}, low = own - 1 iff the offset is negative.
        // Corner order: the eight (node, weight) pairs are the same whatever order they are enumerated in, and
        // every product keeps the reference's operand order.  Three lane bits swap which of an axis's two nodes
        // is visited first, so rays a quarter cell apart that share all 8 target nodes hit different nodes in any
        // one ds_add_f64 instead of serialising on one address: a patch row is lanes 8r..8r+7 and with 4 rays
        // per zone the 16 lanes of rows 0-3 x columns 0-3 share a cell; bits 0 and 1 (column) and bit 3 (row)
        // give those 16 lanes all 8 orders, two lanes each.
        const double ox = (fx - fcx) - 0.5, oy = (fy - fcy) - 0.5, oz = (fz - fcz) - 0.5;   // :319-321
        // Node indices.  The offsets are xtemp - thisx - 0.5 with |xtemp - thisx| < 0.5001, i.e. negative except in
        // a 1e-4-wide sliver: when they are negative on every axis of every live lane (a ballot of the three sign
        // compares: ~98 % of the wave-steps) the low corner is the own node minus one and the first-visited node
        // depends on the lane's flip bits only.
        const bool ngx = ox < 0, ngy = oy < 0, ngz = oz < 0;
        const bool all_negative = (live & ~(CBET_BALLOT(ngx) & CBET_BALLOT(ngy) & CBET_BALLOT(ngz))) == 0ull;
        // the pending sums leave for the nodes they belong to before those are replaced: a lane whose cell changed (the
        // low corner is a function of the cell while the offsets are negative), every lane around the rare branch
        if constexpr (ACC) {
            const unsigned long long fl_m = (all_negative && !p_odd) ? (live & moved_m) : live;
            if (fl_m != 0ull) {   // scalar branch (taken in 99 % of the wave-steps: some lane always moves on)
                if (miss_m == 0ull) flush_sums(fl_m, 0ull);      // (the usual case without the mask arithmetic)
                else flush_sums(fl_m & ~miss_m, fl_m & miss_m);
            }
            restart_m = fl_m;
            p_odd = !all_negative;
        }
        {
            // The two factors of an axis are d = 1 - |o| and 1 - d (:329-336); which of them comes first is the lane's
            // flip bit.  Without selects: F0 = +-(|o| - h), h = 1.0 with the sign flipped for a flipped lane (exactly
            // d), h = 0.0 otherwise (|o| itself: the reference's 1 - (1 - |o|) up to 1.1e-16), F1 = 1 - F0 (exactly
            // 1 - d, or exactly d).  Three instructions and one conversion per axis instead of six.
            auto pair = [](double o, int flip, double &f0, double &f1) {
                const double g = fabs(o) - (double)flip;
                f0 = __hiloint2double(__double2hiint(g) ^ (flip << 31), __double2loint(g));
                f1 = 1.0 - f0;
            };
            pair(ox, pfx, Fx0, Fx1);
            pair(oy, pfy, Fy0, Fy1);
            pair(oz, pfz, Fz0, Fz1);
        }
        // the lane's low corner (haloed) relative to box A's origin: what the common case tests; the corner itself is
        // rebuilt from it where a box has to move (as a value of its own it costs three copies per step)
        int rx, ry, rz;
        if (all_negative) {   // scalar branch
            rx = s.ci - oA.x;
            ry = s.cj - oA.y;
            rz = s.ck - oA.z;
            asm("" : "+v"(rx), "+v"(ry), "+v"(rz));   // (or the compiler merges the two branches' subtractions back into copies + one)
            X0 = xad1(pfx, s.ci); X1 = s.ci + pfx;      // (own + 1 - flip, own + flip)
            Y0 = xad1(pfy, s.cj); Y1 = s.cj + pfy;
            Z0 = xad1(pfz, s.ck); Z1 = s.ck + pfz;
        } else {
            const int lx = s.ci + 1 - (ngx ? 1 : 0);
            const int ly = s.cj + 1 - (ngy ? 1 : 0);
            const int lz = s.ck + 1 - (ngz ? 1 : 0);
            rx = lx - oA.x;
            ry = ly - oA.y;
            rz = lz - oA.z;
            // first-visited node: the own node (the high one iff the offset is negative) unless flipped
            const bool hx = ngx != flx, hy = ngy != fly, hz = ngz != flz;
            X0 = lx + (hx ? 1 : 0); X1 = lx + (hx ? 0 : 1);
            Y0 = ly + (hy ? 1 : 0); Y1 = ly + (hy ? 0 : 1);
            Z0 = lz + (hz ? 1 : 0); Z1 = lz + (hz ? 0 : 1);
        }
        if (CBET && CBET_LANES(live)) {   // the gain gathers are memory accesses: live lanes only
            // path length of the step; u_eff = the ray's energy averaged over the step
            double ds = 0.0;
            // (the hooks' arithmetic is the CBET model's, not the reference's: fused multiply-adds are fine here)
            if (gk || CBET >= 2) ds = sqrt_speed(__builtin_fma(s.vz, s.vz, __builtin_fma(s.vy, s.vy, s.vx * s.vx))) * a.dt;
            double u_eff = s.uray;
            if (gk) {
                // K at the eight deposit nodes, weighted like the deposit.  The pairwise tree makes the
                // sum independent of the corner order (the flips swap operands of commutative adds only).
                const int nX0 = __mul24(X0, sXh), nX1 = __mul24(X1, sXh), nY0 = __mul24(Y0, sYh), nY1 = __mul24(Y1, sYh);
                // The two z nodes of an (x, y) column are neighbours in memory: FOUR 16-byte gathers instead of eight 8-byte
                // ones.  The column sums take the z factors by node (lower, upper), so nothing depends on the lane's z flip;
                // the x and y flips swap operands of commutative adds.
                const bool z0_low = Z0 < Z1;
                const int zl = z0_low ? Z0 : Z1;
                const double fz_lo = z0_low ? Fz0 : Fz1, fz_hi = z0_low ? Fz1 : Fz0;
                const gain_pair_t c00 = gain_load2<IDX64>(a, gk, (unsigned)(nX0 + nY0 + zl)), c10 = gain_load2<IDX64>(a, gk, (unsigned)(nX1 + nY0 + zl));
                const gain_pair_t c01 = gain_load2<IDX64>(a, gk, (unsigned)(nX0 + nY1 + zl)), c11 = gain_load2<IDX64>(a, gk, (unsigned)(nX1 + nY1 + zl));
                // Fused multiply-adds, z then x then y: 14 operations for the 23 of the unfused pairwise tree the CPU checker
                // evaluates (a relative 1e-16 per term; the kernels are held to the checker at 1e-9).  The sum does not depend
                // on the corner order beyond that: the flips swap which of two products is the addend.
                const double q00 = __builtin_fma(fz_hi, c00.y, fz_lo * c00.x), q10 = __builtin_fma(fz_hi, c10.y, fz_lo * c10.x);
                const double q01 = __builtin_fma(fz_hi, c01.y, fz_lo * c01.x), q11 = __builtin_fma(fz_hi, c11.y, fz_lo * c11.x);
                const double r0 = __builtin_fma(Fx1, q10, Fx0 * q00), r1 = __builtin_fma(Fx1, q11, Fx0 * q01);
                const double ksum = __builtin_fma(Fy1, r1, Fy0 * r0);
                double x = ksum * ds;
                if (x > a.max_exponent) x = a.max_exponent;
                if (x < -a.max_exponent) x = -a.max_exponent;
                // (dead lanes excluded from the vote: a.max_exponent bounds |x|, not the garbage a dead lane carries)
                const double phi = (CBET_BALLOT(!(fabs(x) < 0.03125)) & live) == 0ull ? phi_small(x) : phi_det(x);
                const double dg = s.uray * (x * phi);
                u_eff = s.uray * phi;
                gained += dg;
                s.uray = s.uray + dg;
            }
            if (CBET == 2) q0 = u_eff * ds;
            if (CBET == 4) {
                q0 = u_eff * ds;
                q1 = u_eff * (s.vx * a.dt);
                q2 = u_eff * (s.vy * a.dt);
                q3 = u_eff * (s.vz * a.dt);
                const int hi = s.ci + 1, hj = s.cj + 1, hk = s.ck + 1;
                own_slot = T::slot_d(hi & T::XM, hj & T::YM, hk & T::ZM) + NSLOT;
                own_node = __mul24(hi, sXh) + __mul24(hj, sYh) + hk;
            }
        }
        // ---- windows ----------------------------------------------------------------------------------
        // Common case, decided with three compares: every live lane's eight target nodes lie inside its home box
        // -- nothing has to move: the lanes deposit into LDS, box A's while box B is idle (tile_off = 0: every path that
        // retires B leaves it so), and none lies outside both boxes.  (The boxes follow on demand: the step in which a
        // lane leaves is the step in which its box is shifted, before anything is deposited.)
        const unsigned long long memA = live & ~hbm;
        unsigned long long out_core = memA & ~(CBET_BALLOT((unsigned)rx <= (unsigned)T::SX) & CBET_BALLOT((unsigned)ry <= (unsigned)T::SY) &
                                               CBET_BALLOT((unsigned)rz <= (unsigned)T::SZ));
        if (b_active != 0) {   // scalar branch
            asm volatile("");   // (a real branch: if-converted, its assignments cost the common path two selects)
            if constexpr (STATS) wc.slabs_bsteps += 1u;
            const int abx = oA.x - oB.x, aby = oA.y - oB.y, abz = oA.z - oB.z;
            out_core |= hbm & ~(CBET_BALLOT((unsigned)(rx + abx) <= (unsigned)TB::SX) &
                                CBET_BALLOT((unsigned)(ry + aby) <= (unsigned)TB::SY) &
                                CBET_BALLOT((unsigned)(rz + abz) <= (unsigned)TB::SZ));
            tile_off = CBET_LANES(hbm) ? T::N : 0;
        }
        // (lanes the last window pass left outside both boxes send the wave through the general arm again, which clears or
        // renews their mark: 1 % of the wave-steps)
        out_core |= miss_m;
        if (out_core != 0ull) {
#ifdef CBET_DIAG_CLOCKS
            unsigned long long dg_t0;
            asm volatile("s_memtime %0" : "=&s"(dg_t0) : : "memory");
#endif
            // The wave-uniform state this arm changes -- box origins, masks, flags -- is worked on in COPIES and written back
            // at the end through commit(): an assembly move TIED to the variable's register, so that the value the arm leaves
            // and the one the common path carries meet in one register (left to the compiler every such variable costs the
            // common path two scalar copies per step, where the arms join and at the loop's back edge).
            const bool alive = CBET_LANES(live);
            int lx = rx + oA.x, ly = ry + oA.y, lz = rz + oA.z;   // (before box A moves)
            asm volatile("" : "+v"(lx), "+v"(ly), "+v"(lz));      // (formed here: the old origin is not kept for them)
            Origin nA = oA, nB = oB;
            unsigned long long n_hbm = hbm, n_miss = 0ull;
            int n_bact = b_active, n_deep, n_toff = tile_off;
            // box A follows the lanes whose home it is
            follow_box<T, NC>(a, tileA, nA, memA, lx, ly, lz, lane, edep, sXh, sYh, wc, NSLOT, a.comp_stride);
            const unsigned long long lost_mask =
                memA & ~(CBET_BALLOT((unsigned)(lx - nA.x) <= (unsigned)T::SX) & CBET_BALLOT((unsigned)(ly - nA.y) <= (unsigned)T::SY) &
                         CBET_BALLOT((unsigned)(lz - nA.z) <= (unsigned)T::SZ));
            if (lost_mask == 0ull && n_bact == 0) {
                // the usual outcome: A moved and holds every live lane again (tile_off = 0 stands, nobody outside)
                n_deep = box_deep_inside<T>(nA, nx, ny, nz) ? 1 : 0;
            } else {
                bool homeB = CBET_LANES(hbm);
                const bool inA = alive && holds<T>(nA, lx, ly, lz);
                const bool lost = alive && !homeB && !inA;
                // lanes that fell out of A look for a home in B, which follows its own lanes only (letting it chase the
                // lost ones as well was measured: more misses, 0.63 % against 0.48 % of the ray-steps); an idle B is
                // re-created around the first lost lane
                if (n_bact != 0) {  // scalar branch
                    follow_box<TB, 1>(a, tileB, nB, hbm, lx, ly, lz, lane, edep, sXh, sYh, wc, 0, 0);
                } else if (lost_mask != 0ull) {
                    const int src = __ffsll((long long)lost_mask) - 1;
                    const int sx = __builtin_amdgcn_readlane(lx, src), sy = __builtin_amdgcn_readlane(ly, src),
                              sz = __builtin_amdgcn_readlane(lz, src);
                    nB.x = sx - (TB::WX / 2 - 1);
                    nB.y = sy - (TB::WY / 2 - 1);
                    nB.z = sz - TB::SZ / 2;
                    n_bact = 1;  // its tile is all zero: zeroed at start and flushed whenever it empties
                }
                const bool inB = alive && holds<TB>(nB, lx, ly, lz);
                homeB = homeB || (lost && inB);
                // a B lane that drifted out of B but back into A goes home
                if (alive && homeB && !inB && inA) homeB = false;
                n_hbm = CBET_BALLOT(alive && homeB);
                if (n_hbm == 0ull) {
                    __builtin_amdgcn_wave_barrier();
                    flush_box<TB, 1>(a, tileB, nB, lane, edep, sXh, sYh, wc, 0, 0);
                    n_bact = 0;
                }
                __builtin_amdgcn_wave_barrier();
                const bool useB = alive && homeB && inB;
                n_toff = useB ? T::N : 0;
                n_miss = live & ~CBET_BALLOT(useB || (alive && !homeB && inA));
                if constexpr (STATS) {
                    if (n_miss != 0ull) wc.steps_miss += 1u;
                    if (ACC && CBET_LANES(n_miss)) ++wc.n_miss;   // ray-steps whose deposit is bound for HBM (ACC: counted here,
                                                                 // the only place a lane can come to lie outside both boxes)
                }
                n_deep = (n_miss == 0ull && box_deep_inside<T>(nA, nx, ny, nz) && (n_bact == 0 || box_deep_inside<TB>(nB, nx, ny, nz))) ? 1 : 0;
            }
            commit(oA.x, nA.x); commit(oA.y, nA.y); commit(oA.z, nA.z);
            commit(oB.x, nB.x); commit(oB.y, nB.y); commit(oB.z, nB.z);
            commit(hbm, n_hbm); commit(miss_m, n_miss);
            commit(b_active, n_bact); commit(deep, n_deep);
            commit_v(tile_off, n_toff);
            // (the write-back paths keep the count in a vector register; the common path's stays scalar this way)
            wc.pend = __builtin_amdgcn_readfirstlane(wc.pend);
#ifdef CBET_DIAG_CLOCKS
            {
                unsigned long long dg_t1;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dg_t1) : "s"(dg_t0) : "memory");
                dg_shift += dg_t1 - dg_t0;
                dg_nshift += 1;
            }
#endif
        } else if (deep == 0) {
            const int d = (box_deep_inside<T>(oA, nx, ny, nz) && (b_active == 0 || box_deep_inside<TB>(oB, nx, ny, nz))) ? 1 : 0;
            commit(deep, d);
        }
        step_tail(tt);
    }
    tot_steps += __popcll(live) * a.nt;      // rays that ran out of steps (:207)

    // every lane's pending sums (or the last step's deposit), then whatever is still in LDS
    if constexpr (PIPE) accumulate();     // (the last step's deposit of the rays that ran out of steps)
    if constexpr (ACC) flush_sums(live & ~miss_m, live & miss_m);
    else deposit_step(live & ~miss_m, live & miss_m);
    __syncthreads();
    flush_box<T, NC>(a, tileA, oA, lane, edep, sXh, sYh, wc, NSLOT, a.comp_stride);
    if (b_active != 0) flush_box<TB, 1>(a, tileB, oB, lane, edep, sXh, sYh, wc, 0, 0);

    if (CBET && a.beam_gain) {  // one fp64 atomic per wave
        double t = gained;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, kWave);
        if (lane == 0 && t != 0.0) atomicAdd(&a.beam_gain[beam], t);
    }
#ifdef CBET_DIAG_CLOCKS
    if (lane == 0) {   // the diagnostic build reuses four slots: record-wait cycles, shift-path cycles, shift entries, wave cycles
        unsigned long long dg_t_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dg_t_end) : : "memory");
        atomicAdd(&a.counters[kCntSteps], (unsigned long long)tot_steps);
        atomicAdd(&a.counters[kCntWaveSteps], (unsigned long long)(wc.steps_miss >> 16));
        atomicAdd(&a.counters[kCntGlobalAtomics], dg_wait);
        atomicAdd(&a.counters[kCntEvictions], dg_shift);
        atomicAdd(&a.counters[kCntWaveStepsWide], dg_nshift);
        atomicAdd(&a.counters[kCntSlabsRetired], dg_t_end - dg_t_start);
        atomicAdd(&a.counters[kCntRays], wc.dg_ret);              // (the ray count gives way to the write-backs' clocks ...)
        atomicAdd(&a.counters[kCntWaveStepsMiss], wc.dg_nret);    // (... and their number)
    }
    return;
#endif
    // counters: one atomic per wave and counter
    if constexpr (STATS) {
        const int tot_at = wave_sum(wc.n_atomics), tot_miss = wave_sum(wc.n_miss);
        if (lane == 0) {
            atomicAdd(&a.counters[kCntGlobalAtomics], (unsigned long long)tot_at);
            atomicAdd(&a.counters[kCntEvictions], (unsigned long long)tot_miss);
            atomicAdd(&a.counters[kCntWaveSteps], (unsigned long long)(wc.steps_miss >> 16));
            atomicAdd(&a.counters[kCntWaveStepsMiss], (unsigned long long)(wc.steps_miss & 0xFFFFu));
            atomicAdd(&a.counters[kCntWaveStepsWide], (unsigned long long)(wc.slabs_bsteps & 0xFFFFu));
            atomicAdd(&a.counters[kCntSlabsRetired], (unsigned long long)(wc.slabs_bsteps >> 16));
        }
    }
    if (lane == 0) {
        atomicAdd(&a.counters[kCntSteps], (unsigned long long)tot_steps);
        atomicAdd(&a.counters[kCntRays], (unsigned long long)tot_rays);
    }
}

}  // namespace

hipError_t launch_trace_window(const TraceArgs &a, bool force_idx64, hipStream_t stream)
{
    const long waves = a.item_count;
    if (waves <= 0) return hipSuccess;
    const dim3 grid((unsigned)waves), block(kWave);
    // GENERIC is needed for bookkeeping mode, for step-record tables beyond 2^32 bytes (more than 2^27 nodes), for deposit
    // grids of >= 2^32 bytes (a beam's, with the caller's row pitch) and, with the CBET hooks, for gain grids of that size
    // (32-bit byte offsets otherwise)
    const unsigned long long grid_bytes = 8ull * (unsigned long long)a.sXh * (unsigned long long)(a.nx + 2);
    const bool generic = force_idx64 || (a.gain && 8ull * (unsigned long long)a.hsize >= (1ull << 32)) || a.absorption != 1 ||
                         grid_bytes >= (1ull << 32) ||
                         sizeof(StepRecord) * (unsigned long long)a.nx * a.ny * a.nz > (1ull << 32);
    auto go = [&](auto wz, auto cbet) {
        constexpr int WZ = decltype(wz)::value, CB = decltype(cbet)::value;
        if (a.stats) {      // cbet_params.window_stats: the instantiation that also counts the window diagnostics
            if (generic) hipLaunchKernelGGL((k_trace_window<WZ, true, CB, true>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_window<WZ, false, CB, true>), grid, block, 0, stream, a);
        } else {
            if (generic) hipLaunchKernelGGL((k_trace_window<WZ, true, CB, false>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((k_trace_window<WZ, false, CB, false>), grid, block, 0, stream, a);
        }
    };
    using std::integral_constant;
    if (a.quantity == 1) go(integral_constant<int, 8>{}, integral_constant<int, 4>{});        // the fused four-component field pass (single z-planes: four tiles per wave must fit)
    else if (a.quantity == 2) go(integral_constant<int, 16>{}, integral_constant<int, 2>{});  // the energy field alone: the shipped kernel's windows, energy x path length deposited
    else if (a.gain || a.beam_gain) go(integral_constant<int, 16>{}, integral_constant<int, 1>{});
    else go(integral_constant<int, 16>{}, integral_constant<int, 0>{});
    return hipGetLastError();
}

}  // namespace cbet
