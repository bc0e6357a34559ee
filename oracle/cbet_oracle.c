/*
 * cbet_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see cbet_oracle.h for the rules).
 *
 * A run-time-parameterised restatement of the reference ray integrator.  Arithmetic is written
 * operation by operation in the order the reference evaluates it so that a build without FMA
 * contraction (-ffp-contract=off, the Makefile default) reproduces the reference's fp64 results;
 * the only freedom left is the order in which deposits from different rays are summed.
 *
 * Citations are into /root/reference/.
 */
#include "cbet_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Physical constants, def.cuh:61-69, 78, 91, 119, 55-56. */
#define K_C 29979245800.0
#define K_E0 8.85418782e-12
#define K_ME 9.10938356e-31
#define K_EC 1.60217662e-19
#define K_LAMBDA (1.053e-4 / 3.0)
#define K_SIGMA 0.0375
#define K_INTENSITY 1.0e14
#define K_FOCAL 0.1
#define K_BEAM_MIN (-450.0e-4)
#define K_BEAM_MAX (450.0e-4)

static double sq(double v) { return v * v; } /* launch_ray_XZ.cu:9-11 */

static double dmin(double a, double b) { return a < b ? a : b; }

void cbet_oracle_default_config(cbet_oracle_config *cfg, int n)
{
    cfg->nx = cfg->ny = cfg->nz = n;
    cfg->xmin = cfg->ymin = cfg->zmin = -0.13;
    cfg->xmax = cfg->ymax = cfg->zmax = 0.13;
    cfg->nbeams = 60;
    cfg->rays_per_zone = 4;
    cfg->courant_mult = 0.5;
    cfg->absorption = 1;
    cfg->nprofile = 443;
    cfg->max_threads = 120000000;
    cfg->threads_per_block = 256;
}

void cbet_oracle_derive(const cbet_oracle_config *cfg, cbet_oracle_derived *d)
{
    /* def.cuh:39,44,49 */
    d->dx = (cfg->xmax - cfg->xmin) / (cfg->nx - 1);
    d->dy = (cfg->ymax - cfg->ymin) / (cfg->ny - 1);
    d->dz = (cfg->zmax - cfg->zmin) / (cfg->nz - 1);
    /* def.cuh:81 */
    d->dt = cfg->courant_mult * dmin(d->dx, d->dz) / K_C;
    /* def.cuh:83-84: `#if nx >= nz` compares two identifiers the preprocessor reads as 0. */
    d->nt = (int)((1 / cfg->courant_mult) * cfg->nx * 2.0);
    /* launch_ray_XZ.cu:69 ; def.cuh:75-77 (xres==dx, yres==dy, def.cuh:51-52) */
    d->zones_spanned = (int)ceil((K_BEAM_MAX - K_BEAM_MIN) / d->dx);
    d->nrays_x = (int)(cfg->rays_per_zone * ceil((K_BEAM_MAX - K_BEAM_MIN) / d->dx));
    d->nrays_y = (int)(cfg->rays_per_zone * ceil((K_BEAM_MAX - K_BEAM_MIN) / d->dy));
    d->nrays = d->nrays_x * d->nrays_y;
    /* def.cuh:66-69 */
    {
        double freq = K_C / K_LAMBDA;
        d->omega = 2 * M_PI * freq;
        d->ncrit = 1e-6 * (d->omega * d->omega) * K_ME * K_E0 / (K_EC * K_EC);
    }
    /* def.cuh:92 */
    d->uray_mult =
        K_INTENSITY * (cfg->courant_mult) / ((double)(cfg->rays_per_zone * cfg->rays_per_zone));
    /* main.cu:156-159 */
    {
        double grad_const = pow(K_C, 2) / (2.0 * d->ncrit) * d->dt * 0.5;
        d->xconst = grad_const / d->dx;
        d->yconst = grad_const / d->dy;
        d->zconst = grad_const / d->dz;
    }
    /* def.cuh:125-131 ; main.cu:161 */
    {
        long total = (long)d->nrays * cfg->nbeams;
        long nthreads = total < cfg->max_threads ? total : cfg->max_threads;
        d->threads_per_beam = nthreads / cfg->nbeams;
        d->nindices = (int)ceil(d->nrays / (float)(d->threads_per_beam));
        d->grid_y = (int)(d->threads_per_beam / cfg->threads_per_block);
    }
    d->edep_size = ((long)cfg->nx + 2) * ((long)cfg->ny + 2) * ((long)cfg->nz + 2);
}

/* main.cu:24-32 : running sum, not i*step. */
void cbet_oracle_span(double lo, double hi, unsigned len, double *out)
{
    double step = (hi - lo) / (len - 1);
    double acc = lo;
    for (unsigned i = 0; i < len; ++i) {
        out[i] = acc;
        acc += step;
    }
}

/* main.cu:102-110 */
void cbet_oracle_power_table(double *phase_r, double *pow_r)
{
    cbet_oracle_span(0.0, 0.1, CBET_ORACLE_NPHASE, phase_r);
    for (unsigned i = 0; i < CBET_ORACLE_NPHASE; ++i)
        pow_r[i] = exp(-1 * pow(pow((phase_r[i] / K_SIGMA), 2), (5.0 / 2.0)));
}

/* main.cu:121-129 */
void cbet_oracle_beam_trig(const double *beam_norm, int nbeams, double *out4)
{
    for (int b = 0; b < nbeams; ++b) {
        double theta1 = acos(beam_norm[3 * b + 2]);
        double theta2 = atan2(beam_norm[3 * b + 1] * K_FOCAL, beam_norm[3 * b + 0] * K_FOCAL);
        out4[4 * b + 0] = cos(theta1);
        out4[4 * b + 1] = sin(theta1);
        out4[4 * b + 2] = cos(theta2);
        out4[4 * b + 3] = sin(theta2);
    }
}

/* launch_ray_XZ.cu:16-63 : clamped piecewise-linear table lookup, bisection on the abscissa. */
double cbet_oracle_interp(const double *y, const double *x, double xp, int n)
{
    unsigned lo, hi, mid;
    if (x[0] <= x[n - 1]) { /* ascending abscissa, :20-40 */
        if (xp <= x[0]) return y[0];
        if (xp >= x[n - 1]) return y[n - 1];
        lo = 0;
        hi = n - 1;
        mid = (lo + hi) >> 1;
        while (lo < hi - 1) {
            if (x[mid] >= xp)
                hi = mid;
            else
                lo = mid;
            mid = (lo + hi) >> 1;
        }
    } else { /* descending abscissa, :41-61 */
        if (xp >= x[0]) return y[0];
        if (xp <= x[n - 1]) return y[n - 1];
        lo = 0;
        hi = n - 1;
        mid = (lo + hi) >> 1;
        while (lo < hi - 1) {
            if (x[mid] <= xp)
                lo = mid;
            else
                hi = mid;
            mid = (lo + hi) >> 1;
        }
    }
    return y[mid] + (y[mid + 1] - y[mid]) / (x[mid + 1] - x[mid]) * (xp - x[mid]);
}

/* launch_ray_XZ.cu:65-115 */
static int launch_point_d(const cbet_oracle_config *cfg, const cbet_oracle_derived *dd,
                          const double *beam_norm, int beam, int pre_raynum, const double *pow_r,
                          const double *phase_r, double *out)
{
    const cbet_oracle_derived d = *dd;
    const int rpz = cfg->rays_per_zone;

    /* :69-74 thread-ray id -> (rx, ry): 16 consecutive ids tile one launch zone */
    int tile = pre_raynum / (rpz * rpz);
    int within = pre_raynum % (rpz * rpz);
    int ry = tile / d.zones_spanned * rpz + within / rpz;
    int rx = tile % d.zones_spanned * rpz + within % rpz;
    int raynum = ry * d.nrays_x + rx;

    /* :76-92 repeated addition on purpose (":81 in order to agree with CPU") */
    double x0 = K_BEAM_MIN;
    for (int i = 0; i < (raynum % d.nrays_x); i++) x0 += (K_BEAM_MAX - K_BEAM_MIN) / (d.nrays_x - 1);
    x0 += d.dx / 2;
    double y0 = K_BEAM_MIN;
    for (int i = 0; i < (raynum / d.nrays_x); i++) y0 += (K_BEAM_MAX - K_BEAM_MIN) / (d.nrays_y - 1);
    y0 += d.dy / 2;

    double ref = sqrt(sq(x0) + sq(y0)); /* :94 */
    double z0 = K_FOCAL - d.dz / 2;     /* :97 */

    /* :99-111 two rotations */
    double theta1 = acos(beam_norm[beam * 3 + 2]);
    double theta2 = atan2(beam_norm[beam * 3 + 1] * K_FOCAL, K_FOCAL * beam_norm[beam * 3 + 0]);
    double keep = x0;
    x0 = x0 * cos(theta1) + z0 * sin(theta1);
    z0 = z0 * cos(theta1) - keep * sin(theta1);
    double keep2 = x0;
    x0 = x0 * cos(theta2) - y0 * sin(theta2);
    y0 = y0 * cos(theta2) + keep2 * sin(theta2);

    out[0] = x0;
    out[1] = y0;
    out[2] = z0;
    out[3] = d.uray_mult * cbet_oracle_interp(pow_r, phase_r, ref, CBET_ORACLE_NPHASE); /* :113 */
    return ref <= K_BEAM_MAX; /* :114 */
}

int cbet_oracle_launch_point(const cbet_oracle_config *cfg, const double *beam_norm, int beam,
                             int pre_raynum, const double *pow_r, const double *phase_r,
                             double *out)
{
    cbet_oracle_derived d;
    cbet_oracle_derive(cfg, &d);
    return launch_point_d(cfg, &d, beam_norm, beam, pre_raynum, pow_r, phase_r, out);
}

/* launch_ray_XZ.cu:125, 155-158 with main.cu:161's truncating grid.y */
static int id_traced_d(const cbet_oracle_config *cfg, const cbet_oracle_derived *d, int raynum)
{
    if (raynum < 0 || raynum >= d->nrays) return 0;
    long start = raynum % d->threads_per_beam;
    long pass = raynum / d->threads_per_beam;
    return start < (long)d->grid_y * cfg->threads_per_block && pass < d->nindices;
}

int cbet_oracle_id_is_traced(const cbet_oracle_config *cfg, int raynum)
{
    cbet_oracle_derived d;
    cbet_oracle_derive(cfg, &d);
    return id_traced_d(cfg, &d, raynum);
}

/* ---- one ray ------------------------------------------------------------------------------- */

typedef struct ray_ctx {
    const cbet_oracle_config *cfg;
    cbet_oracle_derived d;
    const double *beam_norm, *r, *ne, *te, *pow_r, *phase_r;
    const double *ne3d, *kap3d; /* optional node tables (3-D plasma): replace the radial lookups */
    double *edep;
    int atomic;
    double *path; /* optional KAT recording, 8 doubles per step */
    int path_cap;
    /* CBET extension (UNPINNED: no reference counterpart, see cbet_oracle.h) */
    const double *gain;   /* [nbeams][(n+2)^3] gain coefficient per beam on the deposit grid, 1/cm; NULL = none */
    int quantity;         /* what a step deposits: 0 absorbed energy (reference), 1 ray energy x path length, 2..4 ray energy x displacement x/y/z */
    long grid_stride;     /* doubles between per-beam output grids; 0 = one grid */
    double max_exponent;  /* clamp on |gain * ds| per step */
    double *beam_gain;    /* [nbeams] energy gained through CBET (sum over steps of uray_after - uray_before) */
} ray_ctx;

/* phi(x) = (exp(x) - 1) / x = sum_{n=0}^{17} x^n / (n+1)!  for |x| <= 1, in Horner form from fp64
 * multiplies and adds only (no libm, no FMA), so the HIP kernel evaluates the identical sequence.
 * exp(x) = 1 + x phi(x); phi is also the average of exp(x s) over the step, s in [0,1]. */
static double phi_det(double x)
{
    double p = 1.0 / 6402373705728000.0; /* 1/18! */
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

static void deposit(const ray_ctx *c, long idx, double v)
{
    if (!c->edep) return;
    if (c->atomic) {
#ifdef _OPENMP
#pragma omp atomic
#endif
        c->edep[idx] += v;
    } else {
        c->edep[idx] += v;
    }
}

/* launch_ray_XZ.cu:159-357 for one (beam, thread-ray id).  Returns the number of steps. */
static int trace_one(const ray_ctx *c, int beam, int pre_raynum)
{
    const cbet_oracle_config *cfg = c->cfg;
    const cbet_oracle_derived *d = &c->d;
    const int nx = cfg->nx, ny = cfg->ny, nz = cfg->nz, nr = cfg->nprofile;
    const double xmin = cfg->xmin, ymin = cfg->ymin, zmin = cfg->zmin;
    const double xmax = cfg->xmax, ymax = cfg->ymax, zmax = cfg->zmax;
    const double dx = d->dx, dy = d->dy, dz = d->dz, dt = d->dt;
    const double half = 0.5001; /* :132 */

    double lp[4];
    if (!launch_point_d(cfg, d, c->beam_norm, beam, pre_raynum, c->pow_r, c->phase_r, lp))
        return 0; /* :161,181-183 */
    double px = lp[0], py = lp[1], pz = lp[2], uray = lp[3];
    const double uray0 = uray; /* :160 */

    /* :162-180 first node within 0.5001 cells, scanning upward; 0 when none */
    int ci = 0, cj = 0, ck = 0;
    for (int q = 0; q < nx; ++q)
        if (fabs(q * dx + xmin - px) <= 0.5001 * dx) { ci = q; break; }
    for (int q = 0; q < ny; ++q)
        if (fabs(q * dy + ymin - py) <= 0.5001 * dy) { cj = q; break; }
    for (int q = 0; q < nz; ++q)
        if (fabs(q * dz + zmin - pz) <= 0.5001 * dz) { ck = q; break; }

    /* :186-204 |k| from the dispersion relation at the launch node, direction = -beam normal */
    double rad = sqrt(sq(ci * dx + xmin) + sq(cj * dy + ymin) + sq(ck * dz + zmin));
    /* node value: from the caller's node table when one is given (the 3-D plasma entry of the
     * product), otherwise interpolated at the node's radius as the reference does */
#define NODE_NE(i_, j_, k_, radius_) \
    (c->ne3d ? c->ne3d[((long)(i_) * ny + (j_)) * nz + (k_)] : cbet_oracle_interp(c->ne, c->r, (radius_), nr))
    double ne0 = NODE_NE(ci, cj, ck, rad);
    double w = sqrt((sq(d->omega) - ne0 * 1e6 * sq(K_EC) / ((double)K_ME * K_E0)) / sq(K_C));
    double vx = -1 * c->beam_norm[beam * 3 + 0];
    double vy = -1 * c->beam_norm[beam * 3 + 1];
    double vz = -1 * c->beam_norm[beam * 3 + 2];
    double knorm = sqrt(sq(vx) + sq(vy) + sq(vz));
    vx = sq(K_C) * ((vx / knorm) * w) / d->omega;
    vy = sq(K_C) * ((vy / knorm) * w) / d->omega;
    vz = sq(K_C) * ((vz / knorm) * w) / d->omega;

    const long beam_off = (long)beam * c->grid_stride; /* per-beam output grids (CBET extension) */
    double gained = 0.0;

    int steps = 0;
    for (int tt = 0; tt < d->nt; ++tt) { /* :207 */
        /* :212-238 central-difference neighbours, one-sided at the faces */
        int im = ci - 1, ip = ci + 1, jm = cj - 1, jp = cj + 1, km = ck - 1, kp = ck + 1;
        if (ci == 0) { ip = 2; im = 0; } else if (ci == nx - 1) { ip = nx - 1; im = nx - 3; }
        if (cj == 0) { jp = 2; jm = 0; } else if (cj == ny - 1) { jp = ny - 1; jm = ny - 3; }
        if (ck == 0) { kp = 2; km = 0; } else if (ck == nz - 1) { kp = nz - 1; km = nz - 3; }

        /* :242-250 node coordinates */
        double xp_ = ip * dx + xmin, xm_ = im * dx + xmin, xc_ = ci * dx + xmin;
        double yp_ = jp * dy + ymin, ym_ = jm * dy + ymin, yc_ = cj * dy + ymin;
        double zp_ = kp * dz + zmin, zm_ = km * dz + zmin, zc_ = ck * dz + zmin;

        /* :254-265 density at the six face neighbours of the node the ray sits at before moving */
        double ne_xp = NODE_NE(ip, cj, ck, sqrt(xp_ * xp_ + yc_ * yc_ + zc_ * zc_));
        double ne_xm = NODE_NE(im, cj, ck, sqrt(xm_ * xm_ + yc_ * yc_ + zc_ * zc_));
        double ne_yp = NODE_NE(ci, jp, ck, sqrt(xc_ * xc_ + yp_ * yp_ + zc_ * zc_));
        double ne_ym = NODE_NE(ci, jm, ck, sqrt(xc_ * xc_ + ym_ * ym_ + zc_ * zc_));
        double ne_zp = NODE_NE(ci, cj, kp, sqrt(xc_ * xc_ + yc_ * yc_ + zp_ * zp_));
        double ne_zm = NODE_NE(ci, cj, km, sqrt(xc_ * xc_ + yc_ * yc_ + zm_ * zm_));

        /* :268-273 kick, then drift */
        vx -= d->xconst * (ne_xp - ne_xm);
        vy -= d->yconst * (ne_yp - ne_ym);
        vz -= d->zconst * (ne_zp - ne_zm);
        px += vx * dt;
        py += vy * dt;
        pz += vz * dt;

        /* :276-278 position in cell units */
        double fx = (px - xmin) * (1 / dx);
        double fy = (py - ymin) * (1 / dy);
        double fz = (pz - zmin) * (1 / dz);

        /* :282-292 nearest-node update; the loop's lower bound follows the index as it changes,
         * the lowest matching candidate wins */
        {
            int q = (nx - 1 < ci + 1) ? nx - 1 : ci + 1;
            while (q >= ((0 > ci - 1) ? 0 : ci - 1)) {
                ci = (fabs(q - fx) < half) ? q : ci;
                --q;
            }
            q = (ny - 1 < cj + 1) ? ny - 1 : cj + 1;
            while (q >= ((0 > cj - 1) ? 0 : cj - 1)) {
                cj = (fabs(q - fy) < half) ? q : cj;
                --q;
            }
            q = (nz - 1 < ck + 1) ? nz - 1 : ck + 1;
            while (q >= ((0 > ck - 1) ? 0 : ck - 1)) {
                ck = (fabs(q - fz) < half) ? q : ck;
                --q;
            }
        }

        /* CBET extension (unpinned), part 1: the step's path length; u_eff = the ray's energy averaged
         * over the step (= the arriving energy when no gain acts) is what the field passes deposit */
        double ds = 0.0;
        if (c->gain || c->quantity == 1) ds = sqrt(vx * vx + vy * vy + vz * vz) * dt;
        double u_eff = uray;

        /* :319-336 offsets from the node and the eight linear weights */
        double ox = fx - ci - 0.5;
        double oy = fy - cj - 0.5;
        double oz = fz - ck - 0.5;
        double dm = 1.0 - fabs(ox);
        double dn = 1.0 - fabs(oy);
        double dl = 1.0 - fabs(oz);
        double a1 = (1.0 - dl) * (1.0 - dn) * (1.0 - dm);
        double a2 = (1.0 - dl) * (1.0 - dn) * dm;
        double a3 = dl * (1.0 - dn) * (1.0 - dm);
        double a4 = dl * (1.0 - dn) * dm;
        double a5 = (1.0 - dl) * dn * (1.0 - dm);
        double a6 = (1.0 - dl) * dn * dm;
        double a7 = dl * dn * (1.0 - dm);
        double a8 = dl * dn * dm;
        int sx = (ox < 0) ? -1 : 1, sy = (oy < 0) ? -1 : 1, sz = (oz < 0) ? -1 : 1; /* :338-339 */
        const long sYh = (long)nz + 2, sXh = ((long)ny + 2) * ((long)nz + 2);
        const long hbase = (long)(ci + 1) * sXh + (long)(cj + 1) * sYh + (ck + 1);

        /* CBET extension (unpinned), part 2: the gain coefficient is gathered from the ray's eight deposit
         * nodes with the deposit weights (so that what the rays gain equals sum_nodes K * field, and the
         * pairwise antisymmetry of K makes the exchange between beams conservative once fields and K
         * are consistent); the ray's energy is multiplied by exp(K ds) = 1 + x phi(x) before absorption */
        if (c->gain) {
            const double *gk = c->gain + (long)beam * d->edep_size;
            const double k12 = a1 * gk[hbase] + a2 * gk[hbase + sx * sXh];
            const double k34 = a3 * gk[hbase + sz] + a4 * gk[hbase + sx * sXh + sz];
            const double k56 = a5 * gk[hbase + sy * sYh] + a6 * gk[hbase + sx * sXh + sy * sYh];
            const double k78 = a7 * gk[hbase + sy * sYh + sz] + a8 * gk[hbase + sx * sXh + sy * sYh + sz];
            double x = ((k12 + k34) + (k56 + k78)) * ds;
            if (x > c->max_exponent) x = c->max_exponent;
            if (x < -c->max_exponent) x = -c->max_exponent;
            const double phi = phi_det(x);
            const double dg = uray * (x * phi);
            u_eff = uray * phi;
            gained += dg;
            uray = uray + dg;
        }

        /* :296-311 inverse-bremsstrahlung absorption at the new node */
        double inc;
        if (c->kap3d) { /* caller-supplied absorption factor per node (= ed/ncrit*nuei*dt, see node_tables) */
            if (cfg->absorption == 1) {
                inc = c->kap3d[((long)ci * ny + cj) * nz + ck] * uray;
                uray -= inc;
            } else {
                inc = uray;
            }
        } else {
            double rho = sqrt(sq(ci * dx + xmin) + sq(cj * dy + ymin) + sq(ck * dz + zmin));
            double ed = cbet_oracle_interp(c->ne, c->r, rho, nr);
            double etemp = cbet_oracle_interp(c->te, c->r, rho, nr);
            double eta = 5.2e-5 * 10.0 / (etemp * sqrt(etemp));
            double nuei = (1e6 * ed * sq(K_EC) / K_ME) * eta;
            if (cfg->absorption == 1) {
                inc = ed / d->ncrit * nuei * dt * uray;
                uray -= inc;
            } else {
                inc = uray;
            }
        }


        if (c->quantity == 1) inc = u_eff * ds;             /* energy x path length, the eight deposit weights */
        else if (c->quantity >= 2) {                        /* energy x displacement, at the ray's own node only */
            inc = u_eff * ((c->quantity == 2 ? vx : (c->quantity == 3 ? vy : vz)) * dt);
            a1 = 1.0;
            a2 = a3 = a4 = a5 = a6 = a7 = a8 = 0.0;
        }

        /* :341-348 with :5-7's index */
        {
            const long sY = sYh, sX = sXh;
            long base = beam_off + hbase;
            deposit(c, base, a1 * inc);
            deposit(c, base + sx * sX, a2 * inc);
            deposit(c, base + sz, a3 * inc);
            deposit(c, base + sx * sX + sz, a4 * inc);
            deposit(c, base + sy * sY, a5 * inc);
            deposit(c, base + sx * sX + sy * sY, a6 * inc);
            deposit(c, base + sy * sY + sz, a7 * inc);
            deposit(c, base + sx * sX + sy * sY + sz, a8 * inc);
        }
        if (c->path && steps < c->path_cap) {
            double *p = c->path + 8 * (long)steps;
            p[0] = px; p[1] = py; p[2] = pz;
            p[3] = ci; p[4] = cj; p[5] = ck;
            p[6] = inc; p[7] = uray;
        }
        ++steps;

        /* :351-356 */
        if (uray <= 0.05 * uray0 || px < (xmin - (dx / 2.0)) || px > (xmax + (dx / 2.0)) ||
            py < (ymin - (dy / 2.0)) || py > (ymax + (dy / 2.0)) || pz < (zmin - (dz / 2.0)) ||
            pz > (zmax + (dz / 2.0)))
            break;
    }
    if (c->beam_gain) {
#ifdef _OPENMP
#pragma omp atomic
#endif
        c->beam_gain[beam] += gained;
    }
    return steps;
}

static void ctx_init(ray_ctx *c, const cbet_oracle_config *cfg, const double *beam_norm,
                     const double *r, const double *ne, const double *te, double *phase_r,
                     double *pow_r, double *edep, int atomic)
{
    memset(c, 0, sizeof(*c));
    c->cfg = cfg;
    cbet_oracle_derive(cfg, &c->d);
    c->beam_norm = beam_norm;
    c->r = r;
    c->ne = ne;
    c->te = te;
    cbet_oracle_power_table(phase_r, pow_r);
    c->phase_r = phase_r;
    c->pow_r = pow_r;
    c->edep = edep;
    c->atomic = atomic;
}

long long cbet_oracle_trace(const cbet_oracle_config *cfg, const double *beam_norm,
                            const double *r_prof, const double *ne_prof, const double *te_prof,
                            int beam_lo, int beam_hi, double *edep, int nthreads,
                            long long *steps_per_beam)
{
    static double phase_r[CBET_ORACLE_NPHASE], pow_r[CBET_ORACLE_NPHASE];
    ray_ctx c;
    ctx_init(&c, cfg, beam_norm, r_prof, ne_prof, te_prof, phase_r, pow_r, edep, nthreads > 1);
    const int nrays = c.d.nrays;
    long long total = 0;
    if (steps_per_beam) memset(steps_per_beam, 0, sizeof(long long) * cfg->nbeams);

    for (int beam = beam_lo; beam < beam_hi; ++beam) {
        long long beam_steps = 0;
        if (nthreads > 1) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads) reduction(+ : beam_steps)
#endif
            for (int id = 0; id < nrays; ++id)
                if (id_traced_d(cfg, &c.d, id)) beam_steps += trace_one(&c, beam, id);
        } else {
            for (int id = 0; id < nrays; ++id) /* the serial CPU ray loop */
                if (id_traced_d(cfg, &c.d, id)) beam_steps += trace_one(&c, beam, id);
        }
        if (steps_per_beam) steps_per_beam[beam] = beam_steps;
        total += beam_steps;
    }
    return total;
}

/* Trace with caller-supplied node tables ne3d / kap3d[nx*ny*nz] instead of radial profiles: the
 * checker for the product's 3-D plasma entry (cbet_trace_nodes).  Same ray loop as above. */
long long cbet_oracle_trace_tables(const cbet_oracle_config *cfg, const double *beam_norm,
                                   const double *ne3d, const double *kap3d, int beam_lo, int beam_hi,
                                   double *edep, int nthreads)
{
    static double phase_r[CBET_ORACLE_NPHASE], pow_r[CBET_ORACLE_NPHASE];
    ray_ctx c;
    ctx_init(&c, cfg, beam_norm, NULL, NULL, NULL, phase_r, pow_r, edep, nthreads > 1);
    c.ne3d = ne3d;
    c.kap3d = kap3d;
    const int nrays = c.d.nrays;
    long long total = 0;
    for (int beam = beam_lo; beam < beam_hi; ++beam) {
        long long beam_steps = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 1 ? nthreads : 1) reduction(+ : beam_steps)
#endif
        for (int id = 0; id < nrays; ++id)
            if (id_traced_d(cfg, &c.d, id)) beam_steps += trace_one(&c, beam, id);
        total += beam_steps;
    }
    return total;
}

long long cbet_oracle_trace_list(const cbet_oracle_config *cfg, const double *beam_norm,
                                 const double *r_prof, const double *ne_prof,
                                 const double *te_prof, long nitems, const int *beams,
                                 const int *raynums, double *edep, int nthreads)
{
    static double phase_r[CBET_ORACLE_NPHASE], pow_r[CBET_ORACLE_NPHASE];
    ray_ctx c;
    ctx_init(&c, cfg, beam_norm, r_prof, ne_prof, te_prof, phase_r, pow_r, edep, nthreads > 1);
    long long total = 0;
    if (nthreads > 1) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads) reduction(+ : total)
#endif
        for (long i = 0; i < nitems; ++i)
            if (id_traced_d(cfg, &c.d, raynums[i])) total += trace_one(&c, beams[i], raynums[i]);
    } else {
        for (long i = 0; i < nitems; ++i)
            if (id_traced_d(cfg, &c.d, raynums[i])) total += trace_one(&c, beams[i], raynums[i]);
    }
    return total;
}

int cbet_oracle_ray_path(const cbet_oracle_config *cfg, const double *beam_norm,
                         const double *r_prof, const double *ne_prof, const double *te_prof,
                         int beam, int raynum, int max_steps, double *path)
{
    double phase_r[CBET_ORACLE_NPHASE], pow_r[CBET_ORACLE_NPHASE];
    ray_ctx c;
    ctx_init(&c, cfg, beam_norm, r_prof, ne_prof, te_prof, phase_r, pow_r, NULL, 0);
    c.path = path;
    c.path_cap = max_steps;
    return trace_one(&c, beam, raynum);
}

/* main.cu:6-22 : recursive "[a,b,...]\n" printer, operator<< default formatting == "%g". */
long long cbet_oracle_write_text(const double *edep, int d0, int d1, int d2, const char *path)
{
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    static char buf[1 << 16];
    setvbuf(f, buf, _IOFBF, sizeof buf);
    long long n = 0;
    n += fprintf(f, "[");
    for (int i = 0; i < d0; ++i) {
        n += fprintf(f, "[");
        for (int j = 0; j < d1; ++j) {
            n += fprintf(f, "[");
            const double *row = edep + ((long)i * d1 + j) * d2;
            for (int k = 0; k < d2; ++k) {
                n += fprintf(f, "%g", row[k]);
                if (k + 1 != d2) n += fprintf(f, ",");
            }
            n += fprintf(f, "]\n");
            if (j + 1 != d1) n += fprintf(f, ",");
        }
        n += fprintf(f, "]\n");
        if (i + 1 != d0) n += fprintf(f, ",");
    }
    n += fprintf(f, "]\n");
    fclose(f);
    return n;
}

void cbet_oracle_node_tables(const cbet_oracle_config *cfg, const double *r_prof,
                             const double *ne_prof, const double *te_prof, double *ne3d,
                             double *kap3d)
{
    cbet_oracle_derived d;
    cbet_oracle_derive(cfg, &d);
    const int nr = cfg->nprofile;
    for (int i = 0; i < cfg->nx; ++i)
        for (int j = 0; j < cfg->ny; ++j)
            for (int k = 0; k < cfg->nz; ++k) {
                /* same expression as launch_ray_XZ.cu:296-305 evaluated at node (i,j,k) */
                double rho = sqrt(sq(i * d.dx + cfg->xmin) + sq(j * d.dy + cfg->ymin) +
                                  sq(k * d.dz + cfg->zmin));
                double ed = cbet_oracle_interp(ne_prof, r_prof, rho, nr);
                double etemp = cbet_oracle_interp(te_prof, r_prof, rho, nr);
                double eta = 5.2e-5 * 10.0 / (etemp * sqrt(etemp));
                double nuei = (1e6 * ed * sq(K_EC) / K_ME) * eta;
                long idx = ((long)i * cfg->ny + j) * cfg->nz + k;
                ne3d[idx] = ed;
                if (kap3d) kap3d[idx] = ed / d.ncrit * nuei * d.dt;
            }
}

/* =================================================================================================
 * CBET extension -- PARITY UNPINNED.  The reference has no cross-beam energy transfer code, only the
 * unused constants of def.cuh:94-114; nothing below can be checked against it.  The model is the
 * steady-state ion-acoustic gain of the 2-D ray-based CBET codes those constants come from, written
 * for per-beam intensity / direction FIELDS on the grid:
 *
 *   dI_i/ds = I_i * K_i,     K_i(node) = sum_{j != i} G_ij I_j,    G_ij = -G_ji
 *   G_ij = gain_const * (ne/ncrit) * (1/iaw) * P(eta_ij) / sqrt(eps),   eps = 1 - ne/ncrit
 *   eta_ij = -(k_j - k_i).u / (|k_j - k_i| cs + 1e-10),  P(eta) = iaw^2 eta / ((eta^2-1)^2 + iaw^2 eta^2)
 *   k_b = (omega/c) sqrt(eps) * D_b/|D_b|,  u = Mach(r) cs r_hat  (all beams share one frequency)
 *
 * This file is the checker of the HIP implementation of that model and nothing more.
 * ================================================================================================= */
void cbet_oracle_gain_default(cbet_oracle_gain_config *g)
{
    g->z_ion = 3.1;          /* def.cuh:100 */
    g->te_ev = 2.0e3;        /* def.cuh:104 */
    g->ti_ev = 1.0e3;        /* def.cuh:106 */
    g->mi_over_me = 10230.0; /* def.cuh:101-102 */
    g->iaw = 0.2;            /* def.cuh:107 */
    g->mach_r0 = 0.04;       /* def.cuh:114 uses an undefined `machnum`: a radial outflow ramp stands in */
    g->mach_0 = 0.4;
    g->mach_r1 = 0.13;
    g->mach_1 = 2.4;
    g->max_exponent = 1.0;
}

void cbet_oracle_gain_constants(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                                double *constant1, double *cs, double *gain_const)
{
    cbet_oracle_derived d;
    cbet_oracle_derive(cfg, &d);
    const double estat = 4.80320427e-10;             /* def.cuh:98 */
    const double kb = 1.3806485279e-16;              /* def.cuh:108 */
    const double te_k = g->te_ev * 11604.5052;       /* def.cuh:103 */
    const double ti_k = g->ti_ev * 11604.5052;       /* def.cuh:105 */
    const double mi_kg = g->mi_over_me * K_ME;       /* def.cuh:102 */
    /* def.cuh:111 */
    const double c1 = (pow(estat, 2)) / (4 * (1.0e3 * K_ME) * K_C * d.omega * kb * te_k * (1 + 3 * ti_k / (g->z_ion * te_k)));
    /* def.cuh:113 */
    const double sound = 1e2 * sqrt(K_EC * (g->z_ion * g->te_ev + 3.0 * g->ti_ev) / mi_kg);
    if (constant1) *constant1 = c1;
    if (cs) *cs = sound;
    if (gain_const) *gain_const = c1 * (8.0 * M_PI * 1.0e7 / K_C); /* |E|^2 = 8 pi 1e7 I / c, I in W/cm^2 */
}

long long cbet_oracle_trace_cbet(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                                 const double *beam_norm, const double *ne3d, const double *kap3d,
                                 const double *gain, int quantity, int per_beam, double *out,
                                 double *beam_gain, int nthreads)
{
    static double phase_r[CBET_ORACLE_NPHASE], pow_r[CBET_ORACLE_NPHASE];
    ray_ctx c;
    ctx_init(&c, cfg, beam_norm, NULL, NULL, NULL, phase_r, pow_r, out, 1);
    c.ne3d = ne3d;
    c.kap3d = kap3d;
    c.gain = gain;
    c.quantity = quantity;
    c.grid_stride = per_beam ? c.d.edep_size : 0;
    c.max_exponent = g->max_exponent;
    c.beam_gain = beam_gain;
    if (beam_gain) memset(beam_gain, 0, sizeof(double) * cfg->nbeams);
    const int nrays = c.d.nrays;
    long long total = 0;
    for (int beam = 0; beam < cfg->nbeams; ++beam) {
        long long beam_steps = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 1 ? nthreads : 1) reduction(+ : beam_steps)
#endif
        for (int id = 0; id < nrays; ++id)
            if (id_traced_d(cfg, &c.d, id)) beam_steps += trace_one(&c, beam, id);
        total += beam_steps;
    }
    return total;
}

/* The same for an explicit list of (beam, ray id) pairs: one rank's share in the sharding tests. */
long long cbet_oracle_trace_cbet_list(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                                      const double *beam_norm, const double *ne3d, const double *kap3d,
                                      const double *gain, int quantity, int per_beam, long nitems,
                                      const int *beams, const int *raynums, double *out,
                                      double *beam_gain, int nthreads)
{
    static double phase_r[CBET_ORACLE_NPHASE], pow_r[CBET_ORACLE_NPHASE];
    ray_ctx c;
    ctx_init(&c, cfg, beam_norm, NULL, NULL, NULL, phase_r, pow_r, out, 1);
    c.ne3d = ne3d;
    c.kap3d = kap3d;
    c.gain = gain;
    c.quantity = quantity;
    c.grid_stride = per_beam ? c.d.edep_size : 0;
    c.max_exponent = g->max_exponent;
    c.beam_gain = beam_gain;
    if (beam_gain) memset(beam_gain, 0, sizeof(double) * cfg->nbeams);
    long long total = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads > 1 ? nthreads : 1) reduction(+ : total)
#endif
    for (long it = 0; it < nitems; ++it) total += trace_one(&c, beams[it], raynums[it]);
    return total;
}

double cbet_oracle_phi(double x) { return phi_det(x); }

#define CBET_ORACLE_MAX_BEAMS 64

void cbet_oracle_gain_field(const cbet_oracle_config *cfg, const cbet_oracle_gain_config *g,
                            const double *fields, const double *ne3d, double relax, double *gain,
                            double *change, int nthreads)
{
    cbet_oracle_derived d;
    cbet_oracle_derive(cfg, &d);
    double cs, gain_const;
    cbet_oracle_gain_constants(cfg, g, NULL, &cs, &gain_const);
    const int nx = cfg->nx, ny = cfg->ny, nz = cfg->nz, nb = cfg->nbeams;
    const long sY = (long)nz + 2, sX = ((long)ny + 2) * ((long)nz + 2);
    const long hsize = d.edep_size;
    const double iaw2 = g->iaw * g->iaw;
    const double k0 = d.omega / K_C;
    double sum_change = 0.0, sum_abs = 0.0;
    if (nb > CBET_ORACLE_MAX_BEAMS) { fprintf(stderr, "cbet_oracle_gain_field: more than 64 beams\n"); abort(); }
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1) reduction(+ : sum_change, sum_abs)
#endif
    for (int hi = 0; hi < nx + 2; ++hi)
        for (int hj = 0; hj < ny + 2; ++hj)
            for (int hk = 0; hk < nz + 2; ++hk) {
                /* deposit-grid cell (hi,hj,hk) takes its plasma state from node (hi-1,hj-1,hk-1), clamped */
                const int i = hi < 1 ? 0 : (hi > nx ? nx - 1 : hi - 1);
                const int j = hj < 1 ? 0 : (hj > ny ? ny - 1 : hj - 1);
                const int k = hk < 1 ? 0 : (hk > nz ? nz - 1 : hk - 1);
                const long node = ((long)i * ny + j) * nz + k;
                const long h = (long)hi * sX + (long)hj * sY + hk;
                double raw[CBET_ORACLE_MAX_BEAMS], kx[CBET_ORACLE_MAX_BEAMS], ky[CBET_ORACLE_MAX_BEAMS],
                    kz[CBET_ORACLE_MAX_BEAMS], in[CBET_ORACLE_MAX_BEAMS];
                int present[CBET_ORACLE_MAX_BEAMS], np = 0;
                for (int b = 0; b < nb; ++b) raw[b] = 0.0;
                const double frac = ne3d[node] / d.ncrit;
                const double eps = 1.0 - frac;
                if (eps > 0.0) {
                    const double rt = sqrt(eps);
                    const double kmag = k0 * rt;
                    const double ds_node = (K_C * rt) * d.dt; /* group speed x dt: turns energy x length into intensity */
                    const double xc = i * d.dx + cfg->xmin, yc = j * d.dy + cfg->ymin, zc = k * d.dz + cfg->zmin;
                    const double rr = sqrt(xc * xc + yc * yc + zc * zc);
                    double t = (rr - g->mach_r0) / (g->mach_r1 - g->mach_r0);
                    if (t < 0.0) t = 0.0;
                    if (t > 1.0) t = 1.0;
                    const double um = (g->mach_0 + (g->mach_1 - g->mach_0) * t) * cs;
                    double ux = 0.0, uy = 0.0, uz = 0.0;
                    if (rr > 0.0) { ux = um * (xc / rr); uy = um * (yc / rr); uz = um * (zc / rr); }
                    const double pref = gain_const * frac * (1.0 / g->iaw) / rt;
                    for (int b = 0; b < nb; ++b) {
                        const double E = fields[(0L * nb + b) * hsize + h];
                        const double ax = fields[(1L * nb + b) * hsize + h];
                        const double ay = fields[(2L * nb + b) * hsize + h];
                        const double az = fields[(3L * nb + b) * hsize + h];
                        const double dn = sqrt(ax * ax + ay * ay + az * az);
                        if (E > 0.0 && dn > 0.0) {
                            in[b] = E / ds_node;
                            kx[b] = kmag * (ax / dn);
                            ky[b] = kmag * (ay / dn);
                            kz[b] = kmag * (az / dn);
                            present[np++] = b;
                        }
                    }
                    for (int pi = 0; pi < np; ++pi) {
                        const int bi = present[pi];
                        double acc = 0.0;
                        for (int pj = 0; pj < np; ++pj) {
                            const int bj = present[pj];
                            if (bj == bi) continue;
                            const double qx = kx[bj] - kx[bi], qy = ky[bj] - ky[bi], qz = kz[bj] - kz[bi];
                            const double kiaw = sqrt(qx * qx + qy * qy + qz * qz);
                            if (kiaw > 0.0) {
                                const double eta = (0.0 - (qx * ux + qy * uy + qz * uz)) / (kiaw * cs + 1e-10);
                                const double e2 = eta * eta;
                                const double P = iaw2 * eta / ((e2 - 1.0) * (e2 - 1.0) + iaw2 * e2);
                                acc += pref * P * in[bj];
                            }
                        }
                        raw[bi] = acc;
                    }
                }
                for (int b = 0; b < nb; ++b) {
                    const double old = gain[(long)b * hsize + h];
                    const double nw = old + relax * (raw[b] - old);
                    gain[(long)b * hsize + h] = nw;
                    sum_change += fabs(nw - old);
                    sum_abs += fabs(nw);
                }
            }
    if (change) { change[0] = sum_change; change[1] = sum_abs; }
}
