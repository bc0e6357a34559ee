/*
 * cbet_mi355x.h -- C ABI of the MI355X-native ray-integrator path.
 *
 * This is the drop-in boundary for the ONE hot path of abowman6/CBET_RayTracing_3D: the
 * launch_ray_XYZ kernel launch (launch_ray_XZ.cu:117-121, launched at main.cu:166-176), the two
 * multi_gpu helpers around it (multi_gpu.cuh:6-7) and the rayTracing() orchestrator that strings
 * them together (main.cu:96-232).  Plain pointers and sizes only; no C++ or torch types.
 * Citations are into /root/reference/.
 *
 * Conventions
 *   - every function that can fail returns an int status: CBET_OK (0) or a negative CBET_E* code;
 *     cbet_last_error() returns a thread-local message.  (The reference returns bool + prints to
 *     cout and its call sites ignore the result, main.cu:136-151; here nothing is fire-and-forget.)
 *   - "device pointer" = memory of the HIP device the call names; streams are hipStream_t passed
 *     as void* (NULL = the device's default stream).
 *   - all arithmetic is fp64; grids are dense C-order arrays.
 */
#ifndef CBET_MI355X_H_
#define CBET_MI355X_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBET_OK 0
#define CBET_EINVAL (-1)   /* bad argument / unsupported parameter combination */
#define CBET_EHIP (-2)     /* a HIP runtime call failed (message has hipGetErrorString) */
#define CBET_ENOMEM (-3)   /* device out of memory (safeGPUAlloc's free<size guard) */
#define CBET_ENODEVICE (-4)/* GPUIndex == -1 / no such device (moveToAndFromGPU's guard) */
#define CBET_ECOMM (-5)    /* RCCL failure in the multi-GPU orchestrator */

#define CBET_NPHASE 2001   /* length of pow_r / phase_r, main.cu:102-103 */
#define CBET_MAX_CBET_BEAMS 64 /* the CBET stage keeps a beam-presence bit mask per wavefront */

/* Kernel formulations selectable at run time (cbet_params.kernel_variant). */
#define CBET_KERNEL_DEFAULT 0        /* the library's best parity-exact kernel */
#define CBET_KERNEL_GLOBAL_ATOMICS 1 /* one ray per lane, 8 global fp64 atomics per step */
#define CBET_KERNEL_LDS_COMBINE 2    /* wave-private tagged LDS write-combining of the deposits */
#define CBET_KERNEL_LDS_WINDOW 3     /* wave-private dense LDS windows that follow the ray bundle (the default) */

#define CBET_BEAMS_BY_GPU (-1)       /* cbet_params.beam_hi: no explicit beam range, use the ngpus rule */

/*
 * Run-time counterpart of def.cuh's compile-time configuration (def.cuh:33-131).  Fill with
 * cbet_params_default() and override fields; every launch validates it.
 */
typedef struct cbet_params {
    int nx, ny, nz;              /* def.cuh:35-46 xyz_size (nodes per axis, >= 3)              */
    double xmin, xmax;           /* def.cuh:37-38                                              */
    double ymin, ymax;           /* def.cuh:42-43                                              */
    double zmin, zmax;           /* def.cuh:47-48                                              */
    int nbeams;                  /* def.cuh:58  rows in beam_norm                              */
    int rays_per_zone;           /* def.cuh:71                                                 */
    double courant_mult;         /* def.cuh:80                                                 */
    int absorption;              /* def.cuh:118 (1: absorb; else bookkeeping mode)             */
    int nprofile;                /* def.cuh:33  nr, rows of the radial ne/Te profile           */
    int max_threads;             /* def.cuh:125 (enters the reference's traced-id rule)        */
    int threads_per_block;       /* def.cuh:127 (enters the reference's traced-id rule)        */
    int ngpus;                   /* def.cuh:116 nGPUs: with beam_hi < 0 (unset, the default),  */
                                 /* launch `b` owns beams [b*(nbeams/ngpus), (b+1)*(nbeams/ngpus)), */
                                 /* as at launch_ray_XZ.cu:123                                 */
    int beam_lo, beam_hi;        /* explicit beam range [lo,hi) overriding the ngpus rule;     */
                                 /* beam_hi = CBET_BEAMS_BY_GPU (-1, what cbet_params_default  */
                                 /* sets) = not given.  An empty range [k,k), [0,0) included,  */
                                 /* is an explicit no-op: a rank that owns no beam traces nothing */
    int shard_index, shard_count;/* ray-bundle sharding inside the beam range: the (beam, patch) */
                                 /* list is cut into shard_count contiguous near-equal parts    */
                                 /* and part shard_index is traced (shard_count<=1: everything) */
    int kernel_variant;          /* CBET_KERNEL_*                                              */
    int force_wide_index;        /* test hook: use the 64-bit node-table indexing path that grids  */
                                 /* with 8*nx*ny*nz >= 2^32 bytes (n > 812) need, at any size      */
    int per_beam_grids;          /* 1: beam-resolved deposition -- edep is nbeams grids of            */
                                 /* (nx+2)(ny+2)(nz+2) doubles and beam b adds into grid b (what a    */
                                 /* cross-beam stage needs: every beam's own field); 0: one grid      */
    int patch_order;             /* 1 = longest first: a beam's patches are listed by descending      */
                                 /* launch radius (outer rays take ~3x the steps of central ones);     */
                                 /* 0 = Morton curve; k >= 2 = k radial rings, Morton inside each (all    */
                                 /* measured within 2 % of each other on one device at 256^3).  Part of   */
                                 /* the geometry a context is created for.                               */
    int grid_beam0, grid_beams;  /* beam-resolved arrays (per_beam_grids output, the field pass's output, the  */
                                 /* gain coefficient) that hold only the grids of beams [grid_beam0,          */
                                 /* grid_beam0 + grid_beams): beam b uses grid b - grid_beam0.  grid_beams = 0 */
                                 /* (default): the arrays hold all nbeams grids.  This is how a rank of the    */
                                 /* slab-owned CBET loop keeps only ITS beams' fields and gain.                */
    int rim_merge;               /* patches on the rim of the beam cross-section hold fewer than 64 live rays:  */
                                 /* their rays are pooled, walked by angle around the beam axis and cut into    */
                                 /* bundles of up to 64 rays with a footprint of at most rim_merge launch zones */
                                 /* (= cells) per axis (default 4: 16 rays at 4 rays per zone; 0 = every bundle */
                                 /* is one 8x8 patch).  The same rays are traced; at 256^3 lane utilisation goes from */
                                 /* 0.907 to 0.957 (1620 -> 1552 bundles per beam).  Part of the geometry a     */
                                 /* context is created for.                                                     */
    int edep_zpitch;             /* 0 (default): `edep` has the reference's dense layout, rows of nz+2 doubles   */
                                 /* (launch_ray_XZ.cu:5-7).  p >= nz+2: the caller's grid has rows of p doubles  */
                                 /* -- node (i,j,k) at (i*(ny+2) + j)*p + k; the entries k >= nz+2 of a row are   */
                                 /* padding the launch never touches.  For callers that own their deposit grid's */
                                 /* layout (tracer.SweepPipeline pads its private grids' rows to whole 64-byte    */
                                 /* lines): how the rows of the grid fall on the memory channels relative to the */
                                 /* record table's decides 10 % of the pass time with dense rows of 258 doubles  */
                                 /* (16.9 ... 18.6 ms from one process to the next) and 2 % with rows of 264      */
                                 /* (DESIGN.md section 4.4).  Plain path only.                                    */
    int window_stats;            /* 0 (default): a launch of the default kernel counts ray_steps and rays_traced  */
                                 /* only.  1: the instantiation that also counts the deposit windows' diagnostics  */
                                 /* (global_atomics, lds_evictions, wave_steps, wave_steps_miss, wave_steps_wide,  */
                                 /* slabs_retired) -- a few scalar instructions per step that a timed launch does  */
                                 /* not need.  The cross-check kernels (variants 1, 2) always count what they count. */
} cbet_params;

/* Quantities the reference derives in def.cuh / main.cu:156-161, evaluated in the same order. */
typedef struct cbet_derived {
    double dx, dy, dz, dt;       /* def.cuh:39,44,49,81                                        */
    int nt;                      /* def.cuh:83-84                                              */
    int zones_spanned;           /* launch_ray_XZ.cu:69                                        */
    int nrays_x, nrays_y, nrays; /* def.cuh:75-77                                              */
    double omega, ncrit;         /* def.cuh:68-69                                              */
    double uray_mult;            /* def.cuh:92                                                 */
    double xconst, yconst, zconst; /* main.cu:156-159 dedx_const, dedy_const, dedz_const       */
    long threads_per_beam;       /* def.cuh:128                                                */
    int nindices;                /* def.cuh:129                                                */
    int grid_y;                  /* main.cu:161                                                */
    long edep_size;              /* def.cuh:131 (nx+2)(ny+2)(nz+2)                             */
    long ntraced_ids;            /* thread-ray ids per beam the launch shape visits            */
    long nlive_rays;             /* of those, rays inside the beam radius (per beam)           */
} cbet_derived;

/* Counters a launch accumulates on the device (read back with cbet_context_counters).  The default kernel fills
 * ray_steps and rays_traced always; the other six (the deposit windows' diagnostics) only when cbet_params.window_stats
 * is set -- they stay 0 otherwise.  The cross-check kernels (variants 1, 2) fill what applies to them. */
typedef struct cbet_counters {
    unsigned long long ray_steps;        /* integrator iterations that reached the deposition  */
    unsigned long long rays_traced;      /* live rays launched                                 */
    unsigned long long global_atomics;   /* fp64 atomics that went to HBM                      */
    unsigned long long lds_evictions;    /* LDS_COMBINE: slots written back before the end;    */
                                         /* LDS_WINDOW: ray-steps that fell outside the window */
    unsigned long long wave_steps;       /* integrator iterations per wavefront (64 lane slots each) */
    unsigned long long wave_steps_miss;  /* LDS_WINDOW: wave-steps in which some lane missed the window */
    unsigned long long wave_steps_wide;  /* LDS_WINDOW: wave-steps in which the second box was live     */
    unsigned long long slabs_retired;    /* LDS_WINDOW: wave-steps in which a box origin moved (planes / */
                                         /* z-bricks written back to HBM)                               */
} cbet_counters;

typedef struct cbet_context cbet_context; /* per-device workspace: node tables, ray list, counters */

/* ---- configuration ------------------------------------------------------------------------ */
const char *cbet_last_error(void);
const char *cbet_version(void);

/* def.cuh defaults at grid size n (n = 100 is the shipped configuration). */
int cbet_params_default(cbet_params *p, int n);
int cbet_derive(const cbet_params *p, cbet_derived *d);

/*
 * The beam-independent launch list, in the order the trace kernel consumes it.  The beam cross
 * section is cut into 8x8-ray patches (cbet_params.patch_order); 64 consecutive entries = one ray bundle = one
 * wavefront: one patch, or rays of the rim patches packed together (cbet_params.rim_merge; those bundles come first).  An entry is the thread-ray id (launch_ray_XZ.cu:125,156) of that ray,
 * or -1 for a hole: a ray the reference launch shape never visits or one that fails init()'s
 * beam-radius test (:94,114).  Work items g are (beam, patch) pairs, beam by beam, patch by patch;
 * shard s of K traces the items [s T / K, (s + 1) T / K) of the T in the list.
 * Writes min(n, cap) entries to out (may be NULL) and n to *count.
 */
int cbet_live_ray_list(const cbet_params *p, int *out, long cap, long *count);

/* omega_beams.h:1-62 : the 60 OMEGA port normals, double[60][3], row-major. */
const double *cbet_omega60_beam_norm(void);
/* main.cu:24-32 + 102-110 : phase_r = span(0,0.1,2001), pow_r = exp(-((r/sigma)^2)^2.5); host arrays. */
int cbet_host_power_table(double *phase_r, double *pow_r);
/* main.cu:121-129 : better_beam_norm, 4 doubles per beam {cos t1, sin t1, cos t2, sin t2}; host arrays. */
int cbet_host_beam_trig(const double *beam_norm, int nbeams, double *bbeam_norm);
/* main.cu:249-260 : read exactly nprofile rows "r value" of a profile file into r[] and v[]. */
int cbet_read_profile(const char *path, int nprofile, double *r, double *v);

/* ---- multi_gpu.cuh:6-7 helpers -------------------------------------------------------------- */
/*
 * safeGPUAlloc (multi_gpu.cpp:3-28): makes `gpu` current (and leaves it current), fails with
 * CBET_ENOMEM when free < size, otherwise hipMalloc.  Returns CBET_OK where the reference
 * returns true.
 */
int cbet_safeGPUAlloc(void **dst, size_t size, int gpu);
/*
 * moveToAndFromGPU (multi_gpu.cpp:44-59): gpu == -1 -> CBET_ENODEVICE; blocking copy with the
 * direction inferred from the pointers (hipMemcpyDefault); the caller's current device is
 * preserved.
 */
int cbet_moveToAndFromGPU(void *dst, void *src, size_t size, int gpu);
/* cudaFree counterpart for memory from cbet_safeGPUAlloc (main.cu:180-187). */
int cbet_gpuFree(void *ptr, int gpu);

/* ---- workspace ------------------------------------------------------------------------------ */
/*
 * Allocates the per-device workspace a launch needs (node tables ne3d/kappa3d of nx*ny*nz doubles
 * each, the compacted live-ray list, counters) so that cbet_launch_ray_XYZ itself never
 * allocates or synchronises (graph-capturable).  The context is bound to `gpu` and to the grid /
 * ray geometry in *p; sharding, beam range and kernel variant may change per launch.
 */
int cbet_context_create(cbet_context **ctx, const cbet_params *p, int gpu);
int cbet_context_destroy(cbet_context *ctx);
/* Synchronises `stream`, copies the counters to *out and, if reset != 0, zeroes them. */
int cbet_context_counters(cbet_context *ctx, void *stream, cbet_counters *out, int reset);
/*
 * Replace the context's launch list (cbet_live_ray_list's layout: 64 entries per bundle, -1 = idle lane) by a
 * regrouping of the SAME rays -- every live ray exactly once, no empty bundle; anything else is CBET_EINVAL.  For a
 * caller that knows how long its rays live (e.g. from a previous pass) and wants bundles of rays that end together;
 * the trace deposits the same sums in a different order.  Synchronises the device.
 */
int cbet_context_set_launch_list(cbet_context *ctx, const int *list, long n);
/*
 * Device pointers of the context's own node tables (for tests / the 3-D plasma entry below).  They are writable:
 * the call marks the tables as modified, so the next launch that uses them (ne3d = kappa3d = NULL) rebuilds the
 * per-node step records it gathers from.  Call it again (or cbet_prepare_step_records) after EVERY later in-place
 * modification -- a launch cannot see a write on its own.
 */
int cbet_context_tables(cbet_context *ctx, double **ne3d, double **kappa3d);

/*
 * Bounds-audit builds only (-DCBET_DEBUG_BOUNDS; tests/test_gpu_bounds_audit.py): number of
 * out-of-range grid atomics / node-table gathers / LDS accumulates the kernels attempted (and
 * skipped) on the current device since the last reset.  Regular builds return CBET_EINVAL.
 */
int cbet_debug_bounds_violations(unsigned long long *out, int reset, void *stream);

/* ---- the hot path ---------------------------------------------------------------------------- */
/*
 * launch_ray_XYZ (launch_ray_XZ.cu:117-121; launch site main.cu:171-174).  Same thirteen
 * arguments, same order and meaning:
 *   b          GPU ordinal of the reference's beam split (see cbet_params.ngpus / beam_lo)
 *   nindices   strided passes per thread (def.cuh:129) -- enters the traced-id rule only
 *   te_data_g, r_data_g, ne_data_g   device, nprofile doubles each (main.cu:149-151)
 *   edep       device, (nx+2)(ny+2)(nz+2) doubles, x-major/z-fastest with a one-cell halo
 *              (launch_ray_XZ.cu:5-7); ACCUMULATED into, never cleared (caller zeroes it).
 *              With params->per_beam_grids: nbeams such grids, beam b's at offset b * grid size.
 *   bbeam_norm device, 4*nbeams doubles from cbet_host_beam_trig; the reference uploads but
 *              ignores it (main.cu:146) -- here it supplies the rotation cos/sin so that launch
 *              points equal the host libm's to the last bit.  NULL: trig evaluated on the device.
 *   beam_norm  device, 3*nbeams doubles
 *   pow_r, phase_r  device, CBET_NPHASE doubles each
 *   xconst, yconst, zconst  main.cu:156-159
 * plus: the run-time parameters, the workspace and the stream.  Work is enqueued on `stream`
 * (tabulate node tables -> trace) and the call returns without synchronising; launch errors are
 * reported through the return value (the reference checks nothing, main.cu:171-175).
 */
int cbet_launch_ray_XYZ(int b, unsigned nindices, double *te_data_g, double *r_data_g,
                        double *ne_data_g, double *edep, double *bbeam_norm, double *beam_norm,
                        double *pow_r, double *phase_r, double xconst, double yconst,
                        double zconst, const cbet_params *p, cbet_context *ctx, void *stream);

/*
 * The two halves of cbet_launch_ray_XYZ, for callers that hold a 3-D plasma state:
 *   cbet_tabulate_plasma : radial profiles -> ctx node tables (values the reference would
 *       interpolate at each node, launch_ray_XZ.cu:254-265, 296-305).
 *   cbet_trace_nodes     : trace using caller-supplied node tables ne3d / kappa3d of nx*ny*nz
 *       doubles (NULL = the context's own), i.e. an arbitrary, not necessarily spherical, plasma.
 *   cbet_prepare_step_records : optional.  The default kernel gathers ONE 32-byte record per node and step
 *       -- the three velocity kicks xconst * (ne(x+1) - ne(x-1)) of launch_ray_XZ.cu:268-270 with the face
 *       rule of :212-238, and kappa3d -- which a launch builds from the tables and xconst / yconst / zconst
 *       when they are not current.  Calling this first moves that pass (0.15 ms at 256^3) out of the launch;
 *       records built from the context's own tables stay valid until the next cbet_tabulate_plasma.
 */
int cbet_tabulate_plasma(cbet_context *ctx, const cbet_params *p, const double *te_data_g,
                         const double *r_data_g, const double *ne_data_g, void *stream);
int cbet_trace_nodes(int b, unsigned nindices, const double *ne3d, const double *kappa3d,
                     double *edep, const double *bbeam_norm, const double *beam_norm,
                     const double *pow_r, const double *phase_r, double xconst, double yconst,
                     double zconst, const cbet_params *p, cbet_context *ctx, void *stream);
int cbet_prepare_step_records(cbet_context *ctx, const cbet_params *p, const double *ne3d,
                              const double *kappa3d, double xconst, double yconst, double zconst,
                              void *stream);

/* ---- orchestrator ---------------------------------------------------------------------------- */
/*
 * rayTracing (main.cu:96-232): host profiles in, host edep (caller-owned, (nx+2)(ny+2)(nz+2)
 * doubles) ADDED into.  Uses devices gpus[0..ngpu) (NULL: 0..ngpu-1), one host thread per device;
 * device g of G traces the contiguous part [T g / G, T (g+1) / G) of the beam-major list of T ray bundles
 * (no beam is lost to nbeams/nGPUs truncation), the per-device grids are combined on the devices by one
 * RCCL reduce-scatter into x-slabs, and every device's host thread downloads ITS slab and adds it into
 * edep (replacing the whole-grid D2H copies and the serial host += loop of main.cu:178-210).
 * beam_norm: host double[nbeams][3] (the reference reads the global table of omega_beams.h);
 * NULL = cbet_omega60_beam_norm().  timers (may be NULL) receives {init, tracing, combining,
 * total} seconds as main.cu:219-231 prints; counters (may be NULL) the summed device counters.
 */
int cbet_ray_tracing(const double *te_profile, const double *r_profile, const double *ne_profile,
                     double *edep, const cbet_params *p, const double *beam_norm, const int *gpus,
                     int ngpu, double *timers, cbet_counters *counters);

/* ---- output stage ------------------------------------------------------ */
/*
 * print(std::cout, edep) under -D PRINT (main.cu:6-22, 353-355): the nested "[a,b,...]\n" text
 * at operator<<'s default 6 significant digits -- the format of the reference's golden file
 * truth_100 (Makefile:14-17).  edep: HOST array [d0][d1][d2].  path NULL or "-" = stdout.
 * Returns the number of bytes written, or a negative CBET_E* code.
 */
long long cbet_write_text(const double *edep, int d0, int d1, int d2, const char *path);
/*
 * The 27-point node average of main.cu:334-349 (commented out in the reference):
 * edepavg[nx][ny][nz] from the haloed HOST array edep[nx+2][ny+2][nz+2], same summation order.
 */
int cbet_edep_average(const double *edep, double *edepavg, int nx, int ny, int nz);
/* The same on the current device: edep and edepavg are DEVICE arrays; enqueued on `stream`, same bits. */
int cbet_edep_average_device(const double *edep, double *edepavg, int nx, int ny, int nz, void *stream);
/* main.cu:321-332 (commented out in the reference): HOST arrays x, y, z of [nx][ny][nz] node coordinates. */
int cbet_node_coordinates(const cbet_params *p, double *x, double *y, double *z);
/*
 * Binary output.  The reference's is save2Hdf5 (main.cu:37-94: /Coordinate_x,y,z and /Edepavg, [nx][ny][nz]
 * little-endian fp64) -- dead code there, and libhdf5 does not exist in this image.  The same HOST arrays
 * can be written as NumPy .npy files (format 1.0, C order, '<f8'): data[shape[0]]...[shape[ndim-1]].
 * Returns the number of bytes written, or a negative CBET_E* code.
 */
long long cbet_write_npy(const double *data, int ndim, const long *shape, const char *path);

/* ---- CBET stage ------------------------------------------------------------- */
/*
 * PARITY UNPINNED.  The reference has no cross-beam energy transfer code -- only the unused constants
 * of def.cuh:94-114 (estat, mach, Z, mi, Te, Ti, iaw, kb, constant1, cs, u_flow) -- so this stage has
 * no reference output to match.  It implements the steady-state ion-acoustic gain of the ray-based
 * CBET codes those constants come from, on per-beam FIELDS of the deposit grid:
 *
 *   dI_i/ds = I_i K_i,   K_i = sum_{j != i} G_ij I_j,   G_ij = -G_ji
 *   G_ij = constant1 (8 pi 1e7 / c) (ne/ncrit) (1/iaw) P(eta_ij) / sqrt(1 - ne/ncrit)
 *   eta_ij = -(k_j - k_i).u / (|k_j - k_i| cs + 1e-10),  P = iaw^2 eta / ((eta^2 - 1)^2 + iaw^2 eta^2)
 *
 * with a radial outflow u = Mach(r) cs r_hat (def.cuh:114 refers to an undefined `machnum`).  It is
 * checked against a CPU restatement of the same model (oracle/, tests/test_gpu_cbet.py), by the exact
 * antisymmetry of the exchange and by energy conservation at the fixed point.
 */
typedef struct cbet_gain_params {
    double z_ion;          /* def.cuh:100   Z = 3.1                                             */
    double te_ev, ti_ev;   /* def.cuh:104, 106                                                   */
    double mi_over_me;     /* def.cuh:101-102  10230                                             */
    double iaw;            /* def.cuh:107   ion-acoustic wave damping nu_ia / omega_s            */
    double mach_r0, mach_0, mach_r1, mach_1;  /* Mach number ramps linearly from mach_0 at radius mach_r0 to mach_1 at mach_r1 (clamped outside) */
    double max_exponent;   /* clamp on |K ds| per ray-step, 0 < max_exponent <= 1                */
    double relax;          /* K <- K + relax (K_raw - K) between passes, 0 < relax <= 1          */
    double tolerance;      /* cbet_cbet_solve stops when sum |dK| / sum |K| falls below it       */
    int max_passes;        /* ... or after this many field passes                                */
    int direction_passes;  /* the first direction_passes (>= 1, default 1) field passes deposit all four fields and
                              build the direction field k; later passes deposit the energy field alone and reuse k --
                              gain changes ray energies, not ray paths, so k of the gain-free first pass is kept      */
    int directions_frozen; /* cbet_gain_field*: 1 = the three direction components of `fields` already hold k (an
                              earlier call normalised them): only the energy component is read and normalised       */
    int reserved_;
} cbet_gain_params;

typedef struct cbet_cbet_report {
    int passes;                      /* field passes run                                        */
    int converged;                   /* 1: tolerance reached                                    */
    double change;                   /* last sum |dK| / sum |K|                                 */
    double imbalance;                /* |sum_b gained_b| / sum_b |gained_b| of the final pass   */
    double beam_gain[CBET_MAX_CBET_BEAMS]; /* energy each beam gained in the final pass          */
    unsigned long long ray_steps;    /* all ray-steps traced, field passes included             */
    unsigned long long ray_steps_final; /* ray-steps of the final (deposition) pass              */
} cbet_cbet_report;

int cbet_gain_params_default(cbet_gain_params *g);
/* def.cuh:111 constant1, def.cuh:113 cs, and gain_const = constant1 * 8 pi 1e7 / c (any may be NULL) */
int cbet_gain_constants(const cbet_params *p, const cbet_gain_params *g, double *constant1, double *cs,
                        double *gain_const);
/*
 * cbet_trace_nodes with the CBET hooks.  gain: device [nbeams][(nx+2)(ny+2)(nz+2)] gain coefficient
 * per beam on the deposit grid (1/cm), or NULL; every ray-step gathers it from its eight deposit
 * nodes with the deposit weights and multiplies the ray's energy by exp(K |v| dt) before absorption.
 * quantity says what a step deposits into `out`:
 *   CBET_DEPOSIT_ENERGY  the absorbed energy (the reference's edep; `out` as for cbet_trace_nodes);
 *   CBET_DEPOSIT_FIELD_ENERGY  the first of those four fields alone (out is [nbeams][(n+2)^3], i.e. the head of a
 *                        fields array): what every field pass after the direction-building ones deposits;
 *   CBET_DEPOSIT_FIELDS  the four per-beam fields the gain needs, in ONE trace: out is
 *       [4][nbeams][(n+2)^3] -- component 0 the step-averaged ray energy x path length, spread over the
 *       eight deposit nodes with the deposit weights; components 1..3 that energy x the step's
 *       displacement along x/y/z, at the ray's own (nearest) node.
 * beam_gain: device [nbeams], ADDED into: energy each beam gained (may be NULL).  Default kernel knobs only.
 */
#define CBET_DEPOSIT_ENERGY 0
#define CBET_DEPOSIT_FIELDS 1
#define CBET_DEPOSIT_FIELD_ENERGY 2
int cbet_trace_cbet(int b, unsigned nindices, const double *ne3d, const double *kappa3d,
                    const double *gain, int quantity, double *out, double *beam_gain,
                    const double *bbeam_norm, const double *beam_norm, const double *pow_r,
                    const double *phase_r, double xconst, double yconst, double zconst,
                    const cbet_params *p, const cbet_gain_params *g, cbet_context *ctx, void *stream);
/*
 * fields: device [4][nbeams][(n+2)^3] = what a CBET_DEPOSIT_FIELDS pass deposited;
 * normalised IN PLACE to (intensity, k_x, k_y, k_z) wherever the beam's rays deposited (intensity 0 where the
 * energy is not positive); with g->directions_frozen the k entries are taken as they are and only the energy
 * entries are read and normalised.  gain: device [nbeams][(n+2)^3], updated to
 * gain + relax (K - gain).  change: device double[2], ADDED into: {sum |new - old|, sum |new|}
 * (may be NULL).  scratch: a SELECTOR, kept in the signature for ABI stability: any non-NULL pointer selects
 * the kernel that evaluates every unordered beam pair once with the cell's beams staged in LDS (it is never
 * dereferenced since round 3 -- pass `gain`); NULL selects the ordered kernel (twice the pair evaluations, every
 * statement one IEEE operation in the CPU checker's order; the pair-once kernel's K agrees with it to ~1e-13
 * of the largest |K|).  ne3d NULL = the context's node table.  Needs nbeams <= CBET_MAX_CBET_BEAMS.
 */
int cbet_gain_field(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                    const cbet_params *p, const cbet_gain_params *g, cbet_context *ctx, void *stream);
/*
 * The same for the x-planes [hx_lo, hx_hi) of the deposit grid only (0 <= hx_lo <= hx_hi <= nx+2): one rank's
 * slab when the gain update is shared between ranks (tracer.cbet_fixed_point_slabs).  The arrays keep their full
 * shape; cells outside the slab are neither read for the update nor written.
 */
int cbet_gain_field_slab(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                         int hx_lo, int hx_hi, const cbet_params *p, const cbet_gain_params *g,
                         cbet_context *ctx, void *stream);
/*
 * cbet_gain_field_slab on SLAB-PACKED arrays: fields [4][nbeams][slab] and gain [nbeams][slab] hold only
 * the planes [hx_lo, hx_hi) of every beam's haloed grid (slab = (hx_hi - hx_lo)(ny+2)(nz+2) doubles per beam and
 * component) -- what one rank of the slab-owned loop stores: all beams over its own x-slab.
 */
int cbet_gain_field_packed(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                         int hx_lo, int hx_hi, const cbet_params *p, const cbet_gain_params *g,
                         cbet_context *ctx, void *stream);
/*
 * The sparse exchange of the slab-owned CBET loop (tracer._Exchanger).  A segment is a 64-byte run of 8 doubles aligned
 * to 8 along z; `segments` is a DEVICE array of nseg {row of the array (beam), index of the run inside one beam's
 * [planes][hy][ceil(hz / 8)] run grid} int pairs.  pack: out[8 i .. 8 i + 8) = the i-th run of `src` (rows are
 * beam_stride doubles apart, a row is [planes][hy][hz]; entries beyond the end of a z-row are written as 0);
 * unpack: the inverse scatter (entries beyond the end of a z-row are dropped).  Stream-ordered, no synchronisation.
 */
int cbet_pack_segments(const double *src, long beam_stride, int hy, int hz, const int *segments, long nseg, double *out,
                       void *stream);
int cbet_unpack_segments(double *dst, long beam_stride, int hy, int hz, const int *segments, long nseg, const double *in,
                         void *stream);
/*
 * Device bytes one rank of the slab-owned CBET loop (tracer.cbet_fixed_point_slabs) needs beside the node tables:
 * its own beams' four field components and gain over the whole grid and all beams' fields and gain over its x-slab.
 * The dense exchanges need NO staging: a beam's part over a slab is contiguous in both layouts, so every message is
 * sent from and received into the arrays themselves (one message per beam, peer and component; all peers of a beam
 * in one grouped send/recv).  `_parts`: the rank holds own_beams beams and own_planes planes -- the slabs are cut by
 * gain-update WORK (the central planes, where the beams cross, are the expensive ones), so the outer ranks hold more
 * planes than (nx+2) / W -- plus staging_doubles of exchange staging (0; the optional sparse exchange packs its runs).
 * cbet_cbet_slab_workspace_bytes(p, W, rank): the same with the equal cut of beams and planes (a lower bound for the outer
 * ranks of a balanced cut).  0 on bad arguments.  (512^3, 60 beams, 8 ranks, equal cut: ~86 GB per rank; every rank
 * holding everything, cbet_cbet_workspace_bytes, would be 391 GB.)
 */
size_t cbet_cbet_slab_workspace_bytes(const cbet_params *p, int world_size, int rank);
size_t cbet_cbet_slab_workspace_bytes_parts(const cbet_params *p, int own_beams, int own_planes, size_t staging_doubles);
/* Bytes of device workspace cbet_cbet_solve needs: 5 nbeams (n+2)^3 doubles (four field components + gain) + a few scalars. */
size_t cbet_cbet_workspace_bytes(const cbet_params *p);
/*
 * The whole iteration on the current device: tabulate the plasma; repeat { field pass with the
 * current gain -> normalise -> new gain } until converged; then one deposition pass with the
 * converged gain ADDED into edep (device, (n+2)^3).  Profiles and tables are device pointers as for
 * cbet_launch_ray_XYZ.  workspace: device memory of cbet_cbet_workspace_bytes() or NULL (allocated
 * and freed inside).  Honours shard_index / shard_count only with shard_count == 1 (the multi-rank
 * loop lives above the C ABI: tracer.cbet_solve all-reduces the fields between passes).
 * Synchronises the stream once per pass (it reads the convergence scalars).
 */
int cbet_cbet_solve(double *te_data_g, double *r_data_g, double *ne_data_g, double *edep,
                    double *bbeam_norm, double *beam_norm, double *pow_r, double *phase_r,
                    const cbet_params *p, const cbet_gain_params *g, void *workspace,
                    cbet_context *ctx, void *stream, cbet_cbet_report *report);

#ifdef __cplusplus
}
#endif
#endif /* CBET_MI355X_H_ */
