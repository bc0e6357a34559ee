/*
 * cbet_oracle.h -- CPU oracle for the CBET_RayTracing_3D ray-integrator hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link, import or call it,
 * and there only as the checker / reported CPU baseline.  The shipped path is the HIP library
 * behind include/cbet_mi355x.h and never routes through this code.
 *
 * What it is: a plain-C restatement, with run-time parameters, of the algorithm in
 *   /root/reference/launch_ray_XZ.cu:5-359  (edep_index, square, interp_cuda, init, launch_ray_XYZ)
 *   /root/reference/main.cu:24-32, 102-110, 121-129, 156-161  (span, pow_r, beam trig, constants)
 *   /root/reference/def.cuh:33-131  (derived constants; compile-time macros there, fields here)
 * Each function cites the lines it follows.
 *
 * PARITY UNPINNED (see DESIGN.md "Oracle").  The reference's only golden file, truth_100
 * (Makefile:14-17), is missing from the mount, its tree holds no other fixture or known-answer
 * test, and the reference cannot be built in this image from its own sources (def.cuh includes
 * cuda_runtime.h, H5Cpp.h and boost/multi_array.hpp; nvcc and boost are absent, and stand-in
 * headers are not an allowed build), so oracle/_ref does not exist.  What the restatement IS
 * checked against (tests/test_oracle_golden.py) are the numbers SURVEY.md section 8(c) recorded
 * when the survey session compiled the reference kernel source as host C++ against stub headers:
 * ray-step counts at 64^3/100^3/256^3, sum/max/nonzero counts, individual cell values and the md5
 * of the 6-significant-digit text dump at 100^3.  Those are strong evidence, not a reference-held
 * pin: nobody has compared anything here with truth_100.
 */
#ifndef CBET_ORACLE_H_
#define CBET_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Run-time counterpart of the compile-time knobs in def.cuh (file:line beside each field). */
typedef struct cbet_oracle_config {
    int nx, ny, nz;            /* def.cuh:35-46  xyz_size                           */
    double xmin, xmax;         /* def.cuh:37-38                                     */
    double ymin, ymax;         /* def.cuh:42-43                                     */
    double zmin, zmax;         /* def.cuh:47-48                                     */
    int nbeams;                /* def.cuh:58                                        */
    int rays_per_zone;         /* def.cuh:71                                        */
    double courant_mult;       /* def.cuh:80                                        */
    int absorption;            /* def.cuh:118                                       */
    int nprofile;              /* def.cuh:33   nr                                   */
    int max_threads;           /* def.cuh:125                                       */
    int threads_per_block;     /* def.cuh:127                                       */
} cbet_oracle_config;

/* Quantities def.cuh / main.cu derive from the knobs. */
typedef struct cbet_oracle_derived {
    double dx, dy, dz;         /* def.cuh:39,44,49                                  */
    double dt;                 /* def.cuh:81                                        */
    int nt;                    /* def.cuh:83-84 (the #if always takes branch one)   */
    int zones_spanned;         /* launch_ray_XZ.cu:69                               */
    int nrays_x, nrays_y;      /* def.cuh:75-76                                     */
    int nrays;                 /* def.cuh:77                                        */
    double omega, ncrit;       /* def.cuh:66-69                                     */
    double uray_mult;          /* def.cuh:92                                        */
    double xconst, yconst, zconst; /* main.cu:156-159                               */
    long threads_per_beam;     /* def.cuh:126,128                                   */
    int nindices;              /* def.cuh:129                                       */
    int grid_y;                /* main.cu:161  threads_per_beam/threads_per_block   */
    long edep_size;            /* def.cuh:131                                       */
} cbet_oracle_derived;

#define CBET_ORACLE_NPHASE 2001   /* main.cu:102-103 */

void cbet_oracle_default_config(cbet_oracle_config *cfg, int n);
void cbet_oracle_derive(const cbet_oracle_config *cfg, cbet_oracle_derived *d);

/* main.cu:24-32 */
void cbet_oracle_span(double lo, double hi, unsigned len, double *out);
/* main.cu:102-110 : phase_r and pow_r, CBET_ORACLE_NPHASE entries each */
void cbet_oracle_power_table(double *phase_r, double *pow_r);
/* main.cu:121-129 : cos/sin of theta1, theta2 per beam, 4 doubles per beam */
void cbet_oracle_beam_trig(const double *beam_norm, int nbeams, double *out4);

/* launch_ray_XZ.cu:16-63 */
double cbet_oracle_interp(const double *y, const double *x, double xp, int n);

/* launch_ray_XZ.cu:65-115. out = {x, y, z, uray}; returns 1 when the ray is inside the beam radius. */
int cbet_oracle_launch_point(const cbet_oracle_config *cfg, const double *beam_norm, int beam,
                             int pre_raynum, const double *pow_r, const double *phase_r,
                             double *out);

/* launch_ray_XZ.cu:155-158 + main.cu:161 : is thread-ray id `raynum` visited by the launch shape? */
int cbet_oracle_id_is_traced(const cbet_oracle_config *cfg, int raynum);

/*
 * Trace every ray id the reference launch would visit for beams [beam_lo, beam_hi), adding the
 * deposits into edep[(nx+2)*(ny+2)*(nz+2)] (launch_ray_XZ.cu:117-359).  nthreads<=1 runs the
 * serial ray loop; otherwise OpenMP over rays with atomic adds.  steps_per_beam (may be NULL)
 * receives nbeams counters.  Returns the number of ray-steps (= deposits / 8).
 */
long long cbet_oracle_trace(const cbet_oracle_config *cfg, const double *beam_norm,
                            const double *r_prof, const double *ne_prof, const double *te_prof,
                            int beam_lo, int beam_hi, double *edep, int nthreads,
                            long long *steps_per_beam);

/* Same ray loop, but node values come from caller-supplied tables ne3d / kap3d[nx*ny*nz] (an
 * arbitrary, not necessarily spherical, plasma) -- the checker for cbet_trace_nodes. */
long long cbet_oracle_trace_tables(const cbet_oracle_config *cfg, const double *beam_norm,
                                   const double *ne3d, const double *kap3d, int beam_lo, int beam_hi,
                                   double *edep, int nthreads);

/* Same, for an explicit list of (beam, ray id) pairs -- used by the sharding tests. */
long long cbet_oracle_trace_list(const cbet_oracle_config *cfg, const double *beam_norm,
                                 const double *r_prof, const double *ne_prof,
                                 const double *te_prof, long nitems, const int *beams,
                                 const int *raynums, double *edep, int nthreads);

/*
 * Single-ray known-answer trace: records, per step, position (3), nearest cell (3 ints as doubles),
 * increment and remaining uray into path[8*step ...].  Returns the number of steps taken
 * (0 for a culled ray).  No deposit is made.
 */
int cbet_oracle_ray_path(const cbet_oracle_config *cfg, const double *beam_norm,
                         const double *r_prof, const double *ne_prof, const double *te_prof,
                         int beam, int raynum, int max_steps, double *path);

/* main.cu:6-22 : the `-D PRINT` text rendering of the (nx+2,ny+2,nz+2) array. Returns bytes written. */
long long cbet_oracle_write_text(const double *edep, int d0, int d1, int d2, const char *path);

/* Node tables a 3-D formulation would gather from (checker for the HIP tabulation kernel):
 * ne3d[i][j][k] = interp(ne, r, |x_ijk|), kap3d = per-node absorption factor of launch_ray_XZ.cu:296-305
 * without the trailing `* uray`. */
void cbet_oracle_node_tables(const cbet_oracle_config *cfg, const double *r_prof,
                             const double *ne_prof, const double *te_prof, double *ne3d,
                             double *kap3d);

/* ---- CBET extension: PARITY UNPINNED -----------------------------------------------------------
 * The reference contains no cross-beam energy transfer code (only the unused constants of
 * def.cuh:94-114), so nothing here can be checked against it.  These functions restate on the CPU the
 * field-based gain model the product implements and serve only as the checker
 * of that HIP implementation. */
typedef struct cbet_oracle_gain_config {
    double z_ion;              /* def.cuh:100 */
    double te_ev, ti_ev;       /* def.cuh:104,106 */
    double mi_over_me;         /* def.cuh:101-102 */
    double iaw;                /* def.cuh:107 */
    double mach_r0, mach_0, mach_r1, mach_1; /* radial outflow: Mach number ramps linearly from mach_0 at r0 to mach_1 at r1 */
    double max_exponent;       /* clamp on |gain * path length| per ray-step */
} cbet_oracle_gain_config;

void cbet_oracle_gain_default(cbet_oracle_gain_config *g);
/* def.cuh:111 constant1, def.cuh:113 cs, and constant1 * 8 pi 1e7 / c */
void cbet_oracle_gain_constants(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                                double *constant1, double *cs, double *gain_const);
/* Node-table ray loop of cbet_oracle_trace_tables with the CBET hooks: every step multiplies the ray
 * energy by exp(K |v| dt), K = gain[beam][(n+2)^3] gathered from the step's eight deposit nodes with the
 * deposit weights (gain may be NULL), and deposits `quantity` into out (one (n+2)^3 grid, or nbeams of
 * them when per_beam): 0 the absorbed energy; 1 the step-averaged ray energy x path length (eight deposit
 * weights); 2..4 that energy x the step's displacement along x/y/z, at the ray's own node only.  beam_gain[nbeams] (may be NULL) receives the energy each beam gained. */
long long cbet_oracle_trace_cbet(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                                 const double *beam_norm, const double *ne3d, const double *kap3d,
                                 const double *gain, int quantity, int per_beam, double *out,
                                 double *beam_gain, int nthreads);
/* The same for an explicit list of (beam, ray id) pairs (one rank's share in the sharding tests). */
long long cbet_oracle_trace_cbet_list(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                                      const double *beam_norm, const double *ne3d, const double *kap3d,
                                      const double *gain, int quantity, int per_beam, long nitems,
                                      const int *beams, const int *raynums, double *out,
                                      double *beam_gain, int nthreads);
/* (exp(x) - 1) / x as the ray loop evaluates it (|x| <= 1) */
double cbet_oracle_phi(double x);
/* fields[4][nbeams][(n+2)^3] -> gain[nbeams][(n+2)^3] <- gain + relax * (raw - gain); change = {sum |new-old|, sum |new|} */
void cbet_oracle_gain_field(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                            const double *fields, const double *ne3d, double relax, double *gain,
                            double *change, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
