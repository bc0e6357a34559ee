// cbet_device.h -- argument blocks shared by the host ABI (cbet_abi.cpp) and the gfx950 kernels
// (cbet_kernels.hip).  Internal; the public boundary is include/cbet_mi355x.h.
#ifndef CBET_DEVICE_H_
#define CBET_DEVICE_H_

#include <hip/hip_runtime_api.h>

#include "cbet_mi355x.h"

namespace cbet {

// Physical constants, /root/reference/def.cuh:61-69, 78, 91, 119, 55-56.
constexpr double kC = 29979245800.0;
constexpr double kE0 = 8.85418782e-12;
constexpr double kMe = 9.10938356e-31;
constexpr double kEc = 1.60217662e-19;
constexpr double kLambda = 1.053e-4 / 3.0;
constexpr double kSigma = 0.0375;
constexpr double kIntensity = 1.0e14;
constexpr double kFocal = 0.1;
constexpr double kBeamMin = -450.0e-4;
constexpr double kBeamMax = 450.0e-4;

constexpr unsigned kEmptyTag = 0xFFFFFFFFu;
constexpr int kWave = 64;

// Device counter slots (unsigned long long each); mirrors cbet_counters.
enum CounterSlot { kCntSteps = 0, kCntRays = 1, kCntGlobalAtomics = 2, kCntEvictions = 3, kCntWaveSteps = 4,
                   kCntWaveStepsMiss = 5, kCntWaveStepsWide = 6, kCntSlabsRetired = 7, kCntSlots = 8 };

// Everything the tabulation kernel needs.
struct TabulateArgs {
    int nx, ny, nz, nprofile;
    double xmin, ymin, zmin;
    double dx, dy, dz, dt;
    double ncrit;
    const double *r, *ne, *te;  // device, nprofile each
    double *ne3d, *kap3d;       // device, nx*ny*nz each
};

// What a ray-step of the shipped integrator reads at its (new) node, as ONE 32-byte record: the three velocity
// kicks of launch_ray_XZ.cu:268-270 -- xconst * (ne(x+1) - ne(x-1)) with the one-sided face rule of :212-238,
// the very products the reference forms, tabulated once per node -- and the absorption coefficient of :296-305
// (kappa3d).  One aligned 32-byte gather per lane and step instead of seven 8-byte gathers from five lines.
struct StepRecord {
    double kx, ky, kz, kap;
};

struct StepTableArgs {
    int nx, ny, nz;
    double xconst, yconst, zconst;          // main.cu:156-159
    const double *ne3d, *kap3d;             // node tables, nx*ny*nz each
    StepRecord *rec;                        // out, nx*ny*nz
};

// Everything the trace kernel needs, passed by value as the kernel argument.
struct TraceArgs {
    // grid (def.cuh:35-53)
    int nx, ny, nz;
    double xmin, ymin, zmin;
    double dx, dy, dz, dt;
    double inv_dx, inv_dy, inv_dz;          // (1/dx) of launch_ray_XZ.cu:276-278
    double fx_hi, fy_hi, fz_hi;             // n - 3: cell-unit positions beyond it are "near a face"
    const double *bounds;                   // device: {xlo,xhi,ylo,yhi,zlo,zhi} = xmin-(dx/2.0) ... launch_ray_XZ.cu:352-354
    double tol_x, tol_y, tol_z;             // 0.5001*dx of launch_ray_XZ.cu:164-176
    double xconst, yconst, zconst;          // main.cu:156-159
    int nt, absorption;
    // launch geometry (launch_ray_XZ.cu:65-115)
    int rpz, zones, nrays_x;
    double z_launch;                        // focal_length - dz/2
    double uray_mult, omega, ncrit;
    const double *xlaunch, *ylaunch;        // running-sum launch coordinates, dx/2 already added
    const int *live;                        // compacted thread-ray ids (beam independent)
    int nlive;
    // work split
    int beam_lo, nbeams_local, bundles_per_beam;
    long total_bundles;
    long first_item, item_count;            // this launch's contiguous share of the (beam, patch) list
    // tables
    const double *ne3d, *kap3d;
    const StepRecord *steprec;              // LDS_WINDOW kernel: per-node step records built from the two tables
    const double *beam_norm, *bbeam_norm, *pow_r, *phase_r;
    double *edep;
    int sYh, sXh;                           // strides of the haloed deposit grid in doubles: a row (nz+2, or cbet_params.edep_zpitch), a plane
    long grid_stride;                       // 0: one grid for all beams; else doubles between per-beam grids
    unsigned long long *counters;
    int stats;                              // cbet_params.window_stats: count the window diagnostics too (LDS_WINDOW kernel)
    // bounds-audit builds (-DCBET_DEBUG_BOUNDS) only; unused otherwise
    unsigned long long *audit_count;        // violations counter
    const double *audit_lo, *audit_hi;      // the grid range a launch may add into
    unsigned long long audit_nodes, audit_hsize;  // entries of a node table / of a beam's haloed gain grid
    // CBET extension (no reference counterpart).  All zero / NULL = the reference path.
    int grid_beam0;                         // beam whose grid comes first in a beam-resolved `edep` / in `gain` (cbet_params.grid_beam0)
    const double *gain;                     // [grid beams][(n+2)^3] gain coefficient on the deposit grid, 1/cm
    long hsize;                             // (nx+2)(ny+2)(nz+2)
    int quantity;                           // 0: deposit the absorbed energy; 1: the four field components (fused field pass); 2: the energy field alone
    long comp_stride;                       // field pass: doubles between the component arrays (nbeams * hsize)
    double max_exponent;                    // clamp on |K ds| per step (<= 1)
    double *beam_gain;                      // [nbeams] energy gained through CBET, or NULL
};

// Field normalisation + gain coefficient kernel (CBET extension).
struct GainArgs {
    int nx, ny, nz, nbeams;
    double xmin, ymin, zmin, dx, dy, dz, dt;
    double ncrit, k0;                       // k0 = omega / c
    double cs, gain_const, iaw;
    double mach_r0, mach_0, mach_r1, mach_1;
    double relax;
    double *fields;                         // [4][nbeams][hsize]: (E, Dx, Dy, Dz) in; (I, kx, ky, kz) out where the beam is present
    const double *ne3d;                     // [nx*ny*nz]
    double *gain;                           // [nbeams][hsize]
    double *scratch;                        // [nbeams][hsize] work array of the symmetric kernel, or NULL (ordered kernel)
    double *change;                         // device {sum |new-old|, sum |new|} accumulators, or NULL
    int hx_lo, hx_hi;                       // planes [hx_lo, hx_hi) of the haloed grid to update (a rank's slab; 0 .. nx+2 = all)
    // storage of fields / gain / scratch: entry of cell h of beam b at [b * bstride + h - store0] (+ component * nbeams * bstride)
    int consume;                            // symmetric kernel: leave the energy entries zeroed instead of normalised (the solve loop's next pass accumulates into them)
    int frozen;                             // the direction entries already hold k (cbet_gain_params.directions_frozen)
    long store0, bstride;                   // whole-grid arrays: 0, hsize; slab-packed arrays: hx_lo * (ny+2)(nz+2), slab entries
};

hipError_t launch_tabulate(const TabulateArgs &a, hipStream_t stream);
hipError_t launch_step_table(const StepTableArgs &a, hipStream_t stream);
hipError_t audit_violations(unsigned long long *out, bool reset, hipStream_t stream);
// variant: CBET_KERNEL_GLOBAL_ATOMICS, _LDS_COMBINE (cbet_kernels.hip) or _LDS_WINDOW (cbet_trace_window.hip)
hipError_t launch_trace(const TraceArgs &a, int variant, bool force_idx64, hipStream_t stream);
hipError_t launch_trace_window(const TraceArgs &a, bool force_idx64, hipStream_t stream);
hipError_t launch_gain_field(const GainArgs &a, hipStream_t stream);
hipError_t launch_pack_segments(const double *src, long beam_stride, int hy, int hz, const int *seg, long nseg, double *out,
                                hipStream_t stream);
hipError_t launch_unpack_segments(double *dst, long beam_stride, int hy, int hz, const int *seg, long nseg, const double *in,
                                  hipStream_t stream);
hipError_t launch_edep_average(const double *edep, double *out, int nx, int ny, int nz, hipStream_t stream);

}  // namespace cbet
#endif
