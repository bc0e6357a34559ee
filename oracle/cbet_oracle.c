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
} ray_ctx;

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

        /* :341-348 with :5-7's index */
        {
            const long sY = (long)nz + 2, sX = ((long)ny + 2) * ((long)nz + 2);
            long base = (long)(ci + 1) * sX + (long)(cj + 1) * sY + (ck + 1);
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
