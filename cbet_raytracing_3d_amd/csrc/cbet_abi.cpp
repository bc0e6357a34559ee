// cbet_abi.cpp -- host side of the C ABI in include/cbet_mi355x.h: parameter derivation, the
// per-device workspace, the multi_gpu.cuh helper counterparts and the launch entry points.
// Citations are into /root/reference/.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "cbet_device.h"
#include "cbet_mi355x.h"
#include "cbet_omega_beams.h"

namespace cbet {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    // HIP keeps the last error sticky until read; a reported failure must not leak into the
    // caller's (or torch's) next hipGetLastError() check.
    if (code == CBET_EHIP || code == CBET_ENODEVICE || code == CBET_ENOMEM) (void)hipGetLastError();
    return code;
}

#define CBET_HIP(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ::cbet::fail(CBET_EHIP, "%s failed: %s", #call, hipGetErrorString(e_));      \
    } while (0)

// Restores the caller's current device on scope exit (moveToAndFromGPU's save/restore,
// multi_gpu.cpp:50-57).
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { (void)hipGetDevice(&saved); }
    ~DeviceGuard() { if (saved >= 0) (void)hipSetDevice(saved); }
};

static int validate(const cbet_params *p)
{
    if (!p) return fail(CBET_EINVAL, "params is NULL");
    if (p->nx < 3 || p->ny < 3 || p->nz < 3) return fail(CBET_EINVAL, "grid needs >= 3 nodes per axis");
    if ((long)(p->nx + 2) * (p->ny + 2) * (p->nz + 2) >= 0x7FFFFFFFL)
        return fail(CBET_EINVAL, "grid too large for 32-bit node tags ((n+2)^3 must be < 2^31)");
    // the kernels build cell and haloed-node indices with 24-bit multiplies (v_mul_i32_i24): both operands of
    // (ci*ny + cj)*nz + ck and of X*(ny+2)(nz+2) + Y*(nz+2) + Z must stay below 2^23 (thin anisotropic grids)
    if ((long)p->nx * p->ny >= (1L << 23) || (long)(p->ny + 2) * (p->nz + 2) >= (1L << 23) || p->nx + 2 >= (1 << 23) ||
        p->nz + 2 >= (1 << 23))
        return fail(CBET_EINVAL, "grid too anisotropic for 24-bit index products (need nx*ny < 2^23 and (ny+2)(nz+2) < 2^23)");
    if (!(p->xmax > p->xmin) || !(p->ymax > p->ymin) || !(p->zmax > p->zmin))
        return fail(CBET_EINVAL, "empty extent");
    if (p->nbeams < 1) return fail(CBET_EINVAL, "nbeams < 1");
    if (p->rays_per_zone < 1 || p->rays_per_zone > 64) return fail(CBET_EINVAL, "rays_per_zone out of range");
    if (!(p->courant_mult > 0)) return fail(CBET_EINVAL, "courant_mult <= 0");
    // k_tabulate stages 3 * nprofile doubles in dynamic LDS; 2048 rows = 48 KB, inside the default limit
    if (p->nprofile < 2 || p->nprofile > 2048) return fail(CBET_EINVAL, "nprofile out of range [2,2048]");
    if (p->max_threads < 1 || p->threads_per_block < 1) return fail(CBET_EINVAL, "bad launch-shape rule");
    if (p->shard_count > 1 && (p->shard_index < 0 || p->shard_index >= p->shard_count))
        return fail(CBET_EINVAL, "shard_index outside [0, shard_count)");
    if (p->rim_merge != 0 && (p->rim_merge < 2 || p->rim_merge > 16))
        return fail(CBET_EINVAL, "rim_merge must be 0 (off) or a footprint of 2 .. 16 launch zones");
    if (p->edep_zpitch != 0 && (p->edep_zpitch < p->nz + 2 || (long)(p->ny + 2) * p->edep_zpitch >= (1L << 23) ||
                                (long)(p->nx + 2) * (p->ny + 2) * p->edep_zpitch >= 0x7FFFFFFFL))
        return fail(CBET_EINVAL, "edep_zpitch must be 0 (dense rows) or a row length >= nz + 2 that keeps the grid below 2^31 entries");
    return CBET_OK;
}

// def.cuh:33-131 and main.cu:156-161, evaluated operation by operation as written there.
static void derive_core(const cbet_params *p, cbet_derived *d)
{
    d->dx = (p->xmax - p->xmin) / (p->nx - 1);
    d->dy = (p->ymax - p->ymin) / (p->ny - 1);
    d->dz = (p->zmax - p->zmin) / (p->nz - 1);
    d->dt = p->courant_mult * std::min(d->dx, d->dz) / kC;           // def.cuh:81
    d->nt = (int)((1 / p->courant_mult) * p->nx * 2.0);               // def.cuh:83-84
    d->zones_spanned = (int)std::ceil((kBeamMax - kBeamMin) / d->dx); // launch_ray_XZ.cu:69
    d->nrays_x = (int)(p->rays_per_zone * std::ceil((kBeamMax - kBeamMin) / d->dx));
    d->nrays_y = (int)(p->rays_per_zone * std::ceil((kBeamMax - kBeamMin) / d->dy));
    d->nrays = d->nrays_x * d->nrays_y;
    const double freq = kC / kLambda;                                 // def.cuh:67
    d->omega = 2 * M_PI * freq;                                       // def.cuh:68
    d->ncrit = 1e-6 * (d->omega * d->omega) * kMe * kE0 / (kEc * kEc);// def.cuh:69
    d->uray_mult = kIntensity * (p->courant_mult) / (double(p->rays_per_zone * p->rays_per_zone));
    const double grad_const = std::pow(kC, 2) / (2.0 * d->ncrit) * d->dt * 0.5;  // main.cu:156
    d->xconst = grad_const / d->dx;
    d->yconst = grad_const / d->dy;
    d->zconst = grad_const / d->dz;
    const long total = (long)d->nrays * p->nbeams;                    // def.cuh:125-129
    const long nthreads = std::min<long>(p->max_threads, total);
    d->threads_per_beam = nthreads / p->nbeams;
    d->nindices = (int)std::ceil(d->nrays / (float)(d->threads_per_beam));
    d->grid_y = (int)(d->threads_per_beam / p->threads_per_block);    // main.cu:161
    d->edep_size = ((long)p->nx + 2) * ((long)p->ny + 2) * ((long)p->nz + 2);
    d->ntraced_ids = 0;
    d->nlive_rays = 0;
}

// launch_ray_XZ.cu:125,155-158 with main.cu:161's truncated grid.y: is thread-ray id visited?
static bool id_is_traced(const cbet_params *p, const cbet_derived *d, int nindices, long id)
{
    const long start = id % d->threads_per_beam, pass = id / d->threads_per_beam;
    return start < (long)d->grid_y * p->threads_per_block && pass < nindices;
}

// launch_ray_XZ.cu:76-92 : launch coordinate by repeated addition, then + d/2.
static std::vector<double> launch_axis(int count, int denom_count, double half_cell)
{
    std::vector<double> t(count);
    double acc = kBeamMin;
    for (int i = 0; i < count; ++i) {
        t[i] = acc + half_cell;
        acc += (kBeamMax - kBeamMin) / (denom_count - 1);
    }
    return t;
}

static unsigned morton2(unsigned x, unsigned y)
{
    auto spread = [](unsigned v) {
        v &= 0xFFFF;
        v = (v | (v << 8)) & 0x00FF00FF;
        v = (v | (v << 4)) & 0x0F0F0F0F;
        v = (v | (v << 2)) & 0x33333333;
        v = (v | (v << 1)) & 0x55555555;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

// The beam-independent launch list.  The beam cross-section (nrays_x x nrays_y rays,
// launch_ray_XZ.cu:69-74) is cut into 8x8-ray patches visited along a Morton curve; each patch is
// one ray bundle = one wavefront, lane = 8*row + column.  An entry is the thread-ray id the
// reference would give that ray (the inverse of :70-74's permutation), or -1 for a hole: a ray
// outside the ray grid, one the reference launch shape never visits (:155-158, main.cu:161), or one
// that fails init()'s beam-radius test (:94,114).  Patches with no live ray are dropped.
static void build_live_list(const cbet_params *p, const cbet_derived *d, int nindices,
                            const std::vector<double> &xl, const std::vector<double> &yl,
                            std::vector<int> &slots, long &ntraced, long &nlive)
{
    const int rpz = p->rays_per_zone, rpz2 = rpz * rpz;
    const int zx = d->zones_spanned;
    const int px = (d->nrays_x + 7) / 8, py = (d->nrays_y + 7) / 8;
    // Visit order of the patches.  patch_order 0: Morton curve (neighbouring patches consecutive).
    // patch_order 1 (default): longest rays first -- rays launched far from the beam axis cross the
    // whole box (~4x the steps of the central rays, which are absorbed early), and a launch is only
    // a few rounds of the chip once the work is sharded 8 ways, so dispatching the long bundles
    // first and the short ones last trims the tail.  Ties (and order 0) fall back to Morton.
    std::vector<std::pair<unsigned long long, int>> order;
    order.reserve((size_t)px * py);
    for (int y = 0; y < py; ++y)
        for (int x = 0; x < px; ++x) {
            unsigned long long key = morton2(x, y);
            if (p->patch_order != 0) {
                const int cx = std::min(d->nrays_x - 1, x * 8 + 4), cy = std::min(d->nrays_y - 1, y * 8 + 4);
                const double r2 = xl[cx] * xl[cx] + yl[cy] * yl[cy];
                const double rmax2 = 2.0 * kBeamMax * kBeamMax * 1.1;
                const double levels = p->patch_order == 1 ? 4095.0 : (double)(p->patch_order - 1);   // >= 2: that many radial rings, Morton inside each
                const unsigned long long ring = (unsigned long long)((1.0 - std::min(1.0, r2 / rmax2)) * levels);  // 0 = outermost
                key |= ring << 32;
            }
            order.emplace_back(key, y * px + x);
        }
    std::sort(order.begin(), order.end());
    slots.clear();
    ntraced = 0;
    nlive = 0;
    // ids the launch shape visits, counted once over the whole ray grid
    for (long id = 0; id < d->nrays; ++id)
        if (id_is_traced(p, d, nindices, id)) ++ntraced;
    struct RimRay {
        double angle;
        int id, rx, ry;
    };
    // cbet_params.rim_merge: patches on the rim of the beam hold fewer than 64 live rays, and those rays cross the whole
    // box -- the longest bundles would run with idle lanes (256^3: 144 of 1620 bundles, lane utilisation 0.907).  The rays
    // of all partial patches are pooled, walked by their angle around the beam axis and cut into bundles of up to 64 rays
    // whose footprint stays within rim_merge launch zones per axis (4 = 16 rays: 76 bundles instead of 144, 0.957).  A
    // ray keeps the lane of its patch position where that lane is free, so rays that share a zone still differ in the
    // lane bits that pick the corner order.  The rim bundles -- the longest rays -- head the list.
    const int merge_w = p->rim_merge > 0 ? std::max(8, p->rim_merge * rpz) : 0;   // in rays, never narrower than a patch
    std::vector<RimRay> pool;
    std::vector<int> full;                // the whole patches, in visit order
    std::vector<int> partial;             // the rim patches as they are (kept if packing them gains nothing)
    std::vector<int> packed;
    for (auto &o : order) {
        const int bx = (o.second % px) * 8, by = (o.second / px) * 8;
        int patch[kWave];
        int alive = 0;
        for (int l = 0; l < kWave; ++l) {
            const int rx = bx + (l & 7), ry = by + (l >> 3);
            patch[l] = -1;
            if (rx >= d->nrays_x || ry >= d->nrays_y) continue;
            const long tile = (long)(ry / rpz) * zx + rx / rpz;           // inverse of :72-73
            const long id = tile * rpz2 + (ry % rpz) * rpz + rx % rpz;    // inverse of :70-71
            if (id >= d->nrays || !id_is_traced(p, d, nindices, id)) continue;
            const double ref = std::sqrt(xl[rx] * xl[rx] + yl[ry] * yl[ry]);
            if (!(ref <= kBeamMax)) continue;
            patch[l] = (int)id;
            ++alive;
        }
        if (!alive) continue;
        nlive += alive;
        if (merge_w > 0 && alive < kWave) {
            partial.insert(partial.end(), patch, patch + kWave);
            for (int l = 0; l < kWave; ++l)
                if (patch[l] >= 0) {
                    const int rx = bx + (l & 7), ry = by + (l >> 3);
                    pool.push_back({std::atan2(ry - 0.5 * (d->nrays_y - 1), rx - 0.5 * (d->nrays_x - 1)), patch[l], rx, ry});
                }
        } else {
            full.insert(full.end(), patch, patch + kWave);
        }
    }
    std::sort(pool.begin(), pool.end(), [](const RimRay &u, const RimRay &v) { return u.angle != v.angle ? u.angle < v.angle : u.id < v.id; });
    for (size_t s0 = 0; s0 < pool.size();) {
        size_t e = s0;
        int x0 = pool[s0].rx, x1 = x0, y0 = pool[s0].ry, y1 = y0;
        while (e < pool.size() && e - s0 < (size_t)kWave) {
            const int nx0 = std::min(x0, pool[e].rx), nx1 = std::max(x1, pool[e].rx), ny0 = std::min(y0, pool[e].ry), ny1 = std::max(y1, pool[e].ry);
            if (nx1 - nx0 + 1 > merge_w || ny1 - ny0 + 1 > merge_w) break;
            x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
            ++e;
        }
        int bundle[kWave];
        for (int l = 0; l < kWave; ++l) bundle[l] = -1;
        std::vector<int> extra;           // rays whose own lane is taken
        for (size_t k = s0; k < e; ++k) {
            const int l = (pool[k].rx & 7) + 8 * (pool[k].ry & 7);
            if (bundle[l] < 0) bundle[l] = pool[k].id;
            else extra.push_back(pool[k].id);
        }
        for (int l = 0, q = 0; l < kWave && q < (int)extra.size(); ++l)
            if (bundle[l] < 0) bundle[l] = extra[q++];
        packed.insert(packed.end(), bundle, bundle + kWave);
        s0 = e;
    }
    const std::vector<int> &rim = packed.size() < partial.size() ? packed : partial;
    slots.insert(slots.end(), rim.begin(), rim.end());
    slots.insert(slots.end(), full.begin(), full.end());
}

}  // namespace cbet

using namespace cbet;

struct cbet_context {
    int gpu = -1;
    cbet_params p{};
    cbet_derived d{};
    double *ne3d = nullptr, *kap3d = nullptr;
    StepRecord *steprec = nullptr;      // per-node step records of the LDS_WINDOW kernel (cbet_device.h)
    // what the records were built from: valid while the context's own tables are unchanged (tables_version)
    unsigned long long tables_version = 0, rec_version = ~0ull;
    const double *rec_ne3d = nullptr, *rec_kap3d = nullptr;
    double rec_const[3] = {0, 0, 0};
    double *xlaunch = nullptr, *ylaunch = nullptr;
    double *bounds = nullptr;  // {xlo,xhi,ylo,yhi,zlo,zhi}
    int *live = nullptr;
    int nlive = 0;  // launch-list slots (64 per bundle, holes included)
    unsigned long long *counters = nullptr;
};

extern "C" {

const char *cbet_last_error(void) { return g_err; }
const char *cbet_version(void) { return "cbet-mi355x 0.1 (gfx950, hip)"; }

int cbet_params_default(cbet_params *p, int n)
{
    if (!p) return fail(CBET_EINVAL, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->nx = p->ny = p->nz = n;
    p->xmin = p->ymin = p->zmin = -0.13;
    p->xmax = p->ymax = p->zmax = 0.13;
    p->nbeams = 60;
    p->rays_per_zone = 4;
    p->courant_mult = 0.5;
    p->absorption = 1;
    p->nprofile = 443;
    p->max_threads = 120000000;
    p->threads_per_block = 256;
    p->ngpus = 1;
    p->beam_lo = 0;
    p->beam_hi = CBET_BEAMS_BY_GPU;
    p->shard_index = 0;
    p->shard_count = 1;
    p->kernel_variant = CBET_KERNEL_DEFAULT;
    p->patch_order = 1;
    p->rim_merge = 4;
    return CBET_OK;
}

int cbet_derive(const cbet_params *p, cbet_derived *d)
{
    if (int rc = validate(p)) return rc;
    if (!d) return fail(CBET_EINVAL, "derived is NULL");
    derive_core(p, d);
    if (d->threads_per_beam < 1) return fail(CBET_EINVAL, "no threads per beam");
    auto xl = launch_axis(d->nrays_x, d->nrays_x, d->dx / 2);
    auto yl = launch_axis(d->nrays_y, d->nrays_y, d->dy / 2);
    std::vector<int> live;
    long ntraced = 0, nlive = 0;
    build_live_list(p, d, d->nindices, xl, yl, live, ntraced, nlive);
    d->ntraced_ids = ntraced;
    d->nlive_rays = nlive;
    return CBET_OK;
}

int cbet_live_ray_list(const cbet_params *p, int *out, long cap, long *count)
{
    if (int rc = validate(p)) return rc;
    if (!count) return fail(CBET_EINVAL, "count is NULL");
    cbet_derived d;
    derive_core(p, &d);
    if (d.threads_per_beam < 1) return fail(CBET_EINVAL, "no threads per beam");
    auto xl = launch_axis(d.nrays_x, d.nrays_x, d.dx / 2);
    auto yl = launch_axis(d.nrays_y, d.nrays_y, d.dy / 2);
    std::vector<int> live;
    long ntraced = 0, nlive = 0;
    build_live_list(p, &d, d.nindices, xl, yl, live, ntraced, nlive);
    *count = (long)live.size();
    if (out)
        for (long i = 0; i < std::min<long>(cap, (long)live.size()); ++i) out[i] = live[i];
    return CBET_OK;
}

const double *cbet_omega60_beam_norm(void) { return &cbet_omega60_ports[0][0]; }

int cbet_host_power_table(double *phase_r, double *pow_r)
{
    if (!phase_r || !pow_r) return fail(CBET_EINVAL, "NULL table");
    // main.cu:24-32 span(0.0, 0.1, 2001): running sum
    const double step = (0.1 - 0.0) / (CBET_NPHASE - 1);
    double acc = 0.0;
    for (unsigned i = 0; i < CBET_NPHASE; ++i) {
        phase_r[i] = acc;
        acc += step;
    }
    for (unsigned i = 0; i < CBET_NPHASE; ++i)  // main.cu:108-110
        pow_r[i] = std::exp(-1 * std::pow(std::pow((phase_r[i] / kSigma), 2), (5.0 / 2.0)));
    return CBET_OK;
}

int cbet_host_beam_trig(const double *beam_norm, int nbeams, double *bbeam_norm)
{
    if (!beam_norm || !bbeam_norm || nbeams < 1) return fail(CBET_EINVAL, "bad beam table");
    for (int b = 0; b < nbeams; ++b) {  // main.cu:122-129
        const double theta1 = std::acos(beam_norm[3 * b + 2]);
        const double theta2 = std::atan2(beam_norm[3 * b + 1] * kFocal, beam_norm[3 * b + 0] * kFocal);
        bbeam_norm[4 * b] = std::cos(theta1);
        bbeam_norm[4 * b + 1] = std::sin(theta1);
        bbeam_norm[4 * b + 2] = std::cos(theta2);
        bbeam_norm[4 * b + 3] = std::sin(theta2);
    }
    return CBET_OK;
}

int cbet_read_profile(const char *path, int nprofile, double *r, double *v)
{
    if (!path || !r || !v || nprofile < 1) return fail(CBET_EINVAL, "bad profile arguments");
    FILE *f = std::fopen(path, "r");
    if (!f) return fail(CBET_EINVAL, "cannot open profile file %s", path);
    for (int i = 0; i < nprofile; ++i) {  // main.cu:251-252: exactly nr rows
        if (std::fscanf(f, "%lf %lf", &r[i], &v[i]) != 2) {
            std::fclose(f);
            return fail(CBET_EINVAL, "profile file %s has fewer than %d rows", path, nprofile);
        }
    }
    std::fclose(f);
    return CBET_OK;
}

// ---- multi_gpu.cpp:3-28 ---------------------------------------------------------------------
int cbet_safeGPUAlloc(void **dst, size_t size, int gpu)
{
    if (!dst) return fail(CBET_EINVAL, "dst is NULL");
    hipError_t e = hipSetDevice(gpu);  // stays current, as in the reference (:7)
    if (e != hipSuccess) return fail(CBET_ENODEVICE, "hipSetDevice(%d): %s", gpu, hipGetErrorString(e));
    size_t free_b = 0, total_b = 0;
    e = hipMemGetInfo(&free_b, &total_b);
    if (e != hipSuccess) return fail(CBET_EHIP, "Error encountered during hipMemGetInfo: %s", hipGetErrorString(e));
    if (free_b < size) return fail(CBET_ENOMEM, "GPU: %d is out of memory", gpu);
    e = hipMalloc(dst, size);
    if (e != hipSuccess) return fail(CBET_EHIP, "Error encountered during hipMalloc: %s", hipGetErrorString(e));
    return CBET_OK;
}

// ---- multi_gpu.cpp:44-59 --------------------------------------------------------------------
int cbet_moveToAndFromGPU(void *dst, void *src, size_t size, int gpu)
{
    if (gpu == -1) return fail(CBET_ENODEVICE, "Attempting to move data that has not been assigned a GPU");
    if (size && (!dst || !src)) return fail(CBET_EINVAL, "NULL pointer");
    DeviceGuard guard;
    hipError_t e = hipSetDevice(gpu);
    if (e != hipSuccess) return fail(CBET_ENODEVICE, "hipSetDevice(%d): %s", gpu, hipGetErrorString(e));
    e = hipMemcpy(dst, src, size, hipMemcpyDefault);
    if (e != hipSuccess) return fail(CBET_EHIP, "Error encountered during hipMemcpy: %s", hipGetErrorString(e));
    return CBET_OK;
}

int cbet_gpuFree(void *ptr, int gpu)
{
    DeviceGuard guard;
    hipError_t e = hipSetDevice(gpu);
    if (e != hipSuccess) return fail(CBET_ENODEVICE, "hipSetDevice(%d): %s", gpu, hipGetErrorString(e));
    CBET_HIP(hipFree(ptr));
    return CBET_OK;
}

// ---- workspace --------------------------------------------------------------------------------
int cbet_context_destroy(cbet_context *ctx)
{
    if (!ctx) return CBET_OK;
    DeviceGuard guard;
    (void)hipSetDevice(ctx->gpu);
    (void)hipFree(ctx->ne3d);
    (void)hipFree(ctx->kap3d);
    (void)hipFree(ctx->steprec);
    (void)hipFree(ctx->xlaunch);
    (void)hipFree(ctx->ylaunch);
    (void)hipFree(ctx->bounds);
    (void)hipFree(ctx->live);
    (void)hipFree(ctx->counters);
    delete ctx;
    return CBET_OK;
}

// A caller that knows how long its rays live (a previous pass's per-ray step counts, a model) may regroup the bundles:
// the same rays, every one exactly once, in any grouping of 64 and any order.  Synchronises the device (a launch may
// still be reading the old list).
int cbet_context_set_launch_list(cbet_context *ctx, const int *list, long n)
{
    if (!ctx || !list) return fail(CBET_EINVAL, "NULL context or list");
    if (n <= 0 || n % kWave != 0) return fail(CBET_EINVAL, "a launch list is a whole number of 64-entry bundles (got %ld entries)", n);
    DeviceGuard guard;
    CBET_HIP(hipSetDevice(ctx->gpu));
    CBET_HIP(hipDeviceSynchronize());
    std::vector<int> cur((size_t)ctx->nlive), want, got;
    if (ctx->nlive) CBET_HIP(hipMemcpy(cur.data(), ctx->live, cur.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int v : cur) if (v >= 0) want.push_back(v);
    for (long i = 0; i < n; ++i) {
        if (list[i] >= 0) got.push_back(list[i]);
        else if (list[i] != -1) return fail(CBET_EINVAL, "launch list entry %ld is %d (a ray id or -1)", i, list[i]);
    }
    for (long b = 0; b < n; b += kWave) {
        bool any = false;
        for (int l = 0; l < kWave; ++l) any = any || list[b + l] >= 0;
        if (!any) return fail(CBET_EINVAL, "bundle %ld of the launch list is empty", b / kWave);
    }
    std::sort(want.begin(), want.end());
    std::sort(got.begin(), got.end());
    if (want != got) return fail(CBET_EINVAL, "the launch list must hold exactly the context's live rays, each once (%zu given, %zu expected)", got.size(), want.size());
    int *fresh = nullptr;
    CBET_HIP(hipMalloc((void **)&fresh, (size_t)n * sizeof(int)));
    hipError_t e = hipMemcpy(fresh, list, (size_t)n * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(fresh); return fail(CBET_EHIP, "hipMemcpy(launch list): %s", hipGetErrorString(e)); }
    (void)hipFree(ctx->live);
    ctx->live = fresh;
    ctx->nlive = (int)n;
    return CBET_OK;
}

int cbet_context_create(cbet_context **out, const cbet_params *p, int gpu)
{
    if (!out) return fail(CBET_EINVAL, "ctx out-pointer is NULL");
    *out = nullptr;
    if (int rc = validate(p)) return rc;
    cbet_derived d;
    derive_core(p, &d);
    if (d.threads_per_beam < 1) return fail(CBET_EINVAL, "no threads per beam");
    auto xl = launch_axis(d.nrays_x, d.nrays_x, d.dx / 2);
    auto yl = launch_axis(d.nrays_y, d.nrays_y, d.dy / 2);
    std::vector<int> live;
    long ntraced = 0, nlive = 0;
    build_live_list(p, &d, d.nindices, xl, yl, live, ntraced, nlive);
    d.ntraced_ids = ntraced;
    d.nlive_rays = nlive;

    DeviceGuard guard;
    hipError_t e = hipSetDevice(gpu);
    if (e != hipSuccess) return fail(CBET_ENODEVICE, "hipSetDevice(%d): %s", gpu, hipGetErrorString(e));
    cbet_context *ctx = new cbet_context;
    ctx->gpu = gpu;
    ctx->p = *p;
    ctx->d = d;
    ctx->nlive = (int)live.size();
    const size_t nodes = (size_t)p->nx * p->ny * p->nz;
    auto bail = [&](hipError_t err, const char *what) {
        cbet_context_destroy(ctx);
        return fail(err == hipErrorOutOfMemory ? CBET_ENOMEM : CBET_EHIP, "%s: %s", what, hipGetErrorString(err));
    };
    if ((e = hipMalloc((void **)&ctx->ne3d, nodes * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(ne3d)");
    if ((e = hipMalloc((void **)&ctx->kap3d, nodes * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(kappa3d)");
    if ((e = hipMalloc((void **)&ctx->steprec, nodes * sizeof(StepRecord))) != hipSuccess) return bail(e, "hipMalloc(step records)");
    if ((e = hipMalloc((void **)&ctx->xlaunch, xl.size() * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(xlaunch)");
    if ((e = hipMalloc((void **)&ctx->ylaunch, yl.size() * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(ylaunch)");
    if ((e = hipMalloc((void **)&ctx->bounds, 6 * sizeof(double))) != hipSuccess) return bail(e, "hipMalloc(bounds)");
    {
        // launch_ray_XZ.cu:352-354: xmin - (dx / 2.0), xmax + (dx / 2.0), ...
        const double hb[6] = {p->xmin - (d.dx / 2.0), p->xmax + (d.dx / 2.0), p->ymin - (d.dy / 2.0),
                              p->ymax + (d.dy / 2.0), p->zmin - (d.dz / 2.0), p->zmax + (d.dz / 2.0)};
        if ((e = hipMemcpy(ctx->bounds, hb, sizeof hb, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(bounds)");
    }
    if ((e = hipMalloc((void **)&ctx->live, std::max<size_t>(1, live.size()) * sizeof(int))) != hipSuccess) return bail(e, "hipMalloc(live)");
    if ((e = hipMalloc((void **)&ctx->counters, kCntSlots * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(counters)");
    if ((e = hipMemcpy(ctx->xlaunch, xl.data(), xl.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(xlaunch)");
    if ((e = hipMemcpy(ctx->ylaunch, yl.data(), yl.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(ylaunch)");
    if (!live.empty() &&
        (e = hipMemcpy(ctx->live, live.data(), live.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(live)");
    if ((e = hipMemset(ctx->counters, 0, kCntSlots * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset(counters)");
    *out = ctx;
    return CBET_OK;
}

int cbet_context_counters(cbet_context *ctx, void *stream, cbet_counters *out, int reset)
{
    if (!ctx || !out) return fail(CBET_EINVAL, "NULL context/counters");
    DeviceGuard guard;
    CBET_HIP(hipSetDevice(ctx->gpu));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long h[kCntSlots];
    CBET_HIP(hipMemcpyAsync(h, ctx->counters, sizeof h, hipMemcpyDeviceToHost, s));
    if (reset) CBET_HIP(hipMemsetAsync(ctx->counters, 0, sizeof h, s));
    CBET_HIP(hipStreamSynchronize(s));
    std::memset(out, 0, sizeof *out);
    out->ray_steps = h[kCntSteps];
    out->rays_traced = h[kCntRays];
    out->global_atomics = h[kCntGlobalAtomics];
    out->lds_evictions = h[kCntEvictions];
    out->wave_steps = h[kCntWaveSteps];
    out->wave_steps_miss = h[kCntWaveStepsMiss];
    out->wave_steps_wide = h[kCntWaveStepsWide];
    out->slabs_retired = h[kCntSlabsRetired];
    return CBET_OK;
}

int cbet_debug_bounds_violations(unsigned long long *out, int reset, void *stream)
{
    if (!out) return fail(CBET_EINVAL, "out is NULL");
    hipError_t e = audit_violations(out, reset != 0, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return fail(CBET_EINVAL, "not a bounds-audit build (compile with -DCBET_DEBUG_BOUNDS)");
    if (e != hipSuccess) return fail(CBET_EHIP, "audit_violations: %s", hipGetErrorString(e));
    return CBET_OK;
}

int cbet_context_tables(cbet_context *ctx, double **ne3d, double **kappa3d)
{
    if (!ctx) return fail(CBET_EINVAL, "NULL context");
    if (ne3d) *ne3d = ctx->ne3d;
    if (kappa3d) *kappa3d = ctx->kap3d;
    ++ctx->tables_version;   // the pointers are writable: records built from the old contents are no longer trusted
    return CBET_OK;
}

// The launch must describe the grid / ray geometry the workspace was sized for.
static int check_geometry(const cbet_context *ctx, const cbet_params *p)
{
    const cbet_params &q = ctx->p;
    if (p->nx != q.nx || p->ny != q.ny || p->nz != q.nz || p->xmin != q.xmin || p->xmax != q.xmax ||
        p->ymin != q.ymin || p->ymax != q.ymax || p->zmin != q.zmin || p->zmax != q.zmax ||
        p->rays_per_zone != q.rays_per_zone || p->nbeams != q.nbeams || p->nprofile != q.nprofile ||
        p->max_threads != q.max_threads || p->threads_per_block != q.threads_per_block ||
        p->courant_mult != q.courant_mult || p->patch_order != q.patch_order || p->rim_merge != q.rim_merge)
        return fail(CBET_EINVAL, "launch parameters do not match the geometry the context was created for");
    return CBET_OK;
}

int cbet_tabulate_plasma(cbet_context *ctx, const cbet_params *p, const double *te_data_g,
                         const double *r_data_g, const double *ne_data_g, void *stream)
{
    if (!ctx) return fail(CBET_EINVAL, "NULL context");
    if (int rc = validate(p)) return rc;
    if (int rc = check_geometry(ctx, p)) return rc;
    if (!te_data_g || !r_data_g || !ne_data_g) return fail(CBET_EINVAL, "NULL profile pointer");
    DeviceGuard guard;
    CBET_HIP(hipSetDevice(ctx->gpu));
    TabulateArgs t{};
    t.nx = p->nx; t.ny = p->ny; t.nz = p->nz; t.nprofile = p->nprofile;
    t.xmin = p->xmin; t.ymin = p->ymin; t.zmin = p->zmin;
    t.dx = ctx->d.dx; t.dy = ctx->d.dy; t.dz = ctx->d.dz; t.dt = ctx->d.dt;
    t.ncrit = ctx->d.ncrit;
    t.r = r_data_g; t.ne = ne_data_g; t.te = te_data_g;
    t.ne3d = ctx->ne3d; t.kap3d = ctx->kap3d;
    CBET_HIP(launch_tabulate(t, (hipStream_t)stream));
    ++ctx->tables_version;   // step records built from the old tables are stale
    return CBET_OK;
}

// Build the per-node step records (cbet_device.h StepRecord) the LDS_WINDOW kernel gathers from: ne3d / kappa3d
// NULL = the context's own tables.  Records built from the context's tables stay valid until the next
// cbet_tabulate_plasma; records built from caller-owned tables are rebuilt by every launch (their contents may
// have changed).
static int step_records(cbet_context *ctx, const cbet_params *p, const double *ne3d, const double *kappa3d,
                        double xconst, double yconst, double zconst, void *stream, bool force)
{
    const bool own = !ne3d && !kappa3d;
    const double *ne = ne3d ? ne3d : ctx->ne3d, *kap = kappa3d ? kappa3d : ctx->kap3d;
    if (!force && own && ctx->rec_version == ctx->tables_version && ctx->rec_ne3d == ne && ctx->rec_kap3d == kap &&
        ctx->rec_const[0] == xconst && ctx->rec_const[1] == yconst && ctx->rec_const[2] == zconst)
        return CBET_OK;
    StepTableArgs t{};
    t.nx = p->nx; t.ny = p->ny; t.nz = p->nz;
    t.xconst = xconst; t.yconst = yconst; t.zconst = zconst;
    t.ne3d = ne; t.kap3d = kap; t.rec = ctx->steprec;
    CBET_HIP(launch_step_table(t, (hipStream_t)stream));
    ctx->rec_version = own ? ctx->tables_version : ~0ull;
    ctx->rec_ne3d = ne; ctx->rec_kap3d = kap;
    ctx->rec_const[0] = xconst; ctx->rec_const[1] = yconst; ctx->rec_const[2] = zconst;
    return CBET_OK;
}

int cbet_prepare_step_records(cbet_context *ctx, const cbet_params *p, const double *ne3d, const double *kappa3d,
                              double xconst, double yconst, double zconst, void *stream)
{
    if (!ctx) return fail(CBET_EINVAL, "NULL context");
    if (int rc = validate(p)) return rc;
    if (int rc = check_geometry(ctx, p)) return rc;
    DeviceGuard guard;
    CBET_HIP(hipSetDevice(ctx->gpu));
    return step_records(ctx, p, ne3d, kappa3d, xconst, yconst, zconst, stream, true);
}

// CBET hooks of a trace launch (all zero: the reference path).
struct CbetHooks {
    const double *gain = nullptr;
    int quantity = 0;
    double *beam_gain = nullptr;
    double max_exponent = 0.0;
};

static int trace_impl(int b, unsigned nindices, const double *ne3d, const double *kappa3d,
                      double *edep, const double *bbeam_norm, const double *beam_norm,
                      const double *pow_r, const double *phase_r, double xconst, double yconst,
                      double zconst, const cbet_params *p, cbet_context *ctx, void *stream,
                      const CbetHooks &hooks)
{
    if (!ctx) return fail(CBET_EINVAL, "NULL context");
    if (int rc = validate(p)) return rc;
    if (int rc = check_geometry(ctx, p)) return rc;
    if (!edep || !beam_norm || !pow_r || !phase_r) return fail(CBET_EINVAL, "NULL device pointer");
    if (nindices == 0) return CBET_OK;  // launch_ray_XZ.cu:155: the ray loop does not run
    if ((int)nindices != ctx->d.nindices)
        return fail(CBET_EINVAL, "nindices=%u but def.cuh:129 gives %d for these parameters", nindices, ctx->d.nindices);

    int beam_lo = p->beam_lo, beam_hi = p->beam_hi;
    // "unset" is beam_hi < 0 (CBET_BEAMS_BY_GPU, the default); every empty range [k,k), [0,0) included, is an
    // explicit no-op (a rank that owns no beam), handled below
    if (beam_hi >= 0 && beam_hi < beam_lo) return fail(CBET_EINVAL, "beam range [%d,%d) is reversed", beam_lo, beam_hi);
    if (beam_hi < 0) {  // launch_ray_XZ.cu:123 with grid.x = nbeams/nGPUs (main.cu:161)
        const int ng = p->ngpus > 0 ? p->ngpus : 1;
        const int per = p->nbeams / ng;
        beam_lo = b * per;
        beam_hi = beam_lo + per;
    }
    if (beam_lo < 0 || beam_hi > p->nbeams || beam_hi < beam_lo)
        return fail(CBET_EINVAL, "beam range [%d,%d) outside [0,%d)", beam_lo, beam_hi, p->nbeams);
    if (beam_hi == beam_lo || ctx->nlive == 0) return CBET_OK;

    int variant = p->kernel_variant;
    if (variant == CBET_KERNEL_DEFAULT) variant = CBET_KERNEL_LDS_WINDOW;
    if (variant != CBET_KERNEL_GLOBAL_ATOMICS && variant != CBET_KERNEL_LDS_COMBINE &&
        variant != CBET_KERNEL_LDS_WINDOW)
        return fail(CBET_EINVAL, "unknown kernel_variant %d", p->kernel_variant);
    const bool cbet_hooks = hooks.gain || hooks.quantity != 0 || hooks.beam_gain;
    if (cbet_hooks && variant != CBET_KERNEL_LDS_WINDOW)
        return fail(CBET_EINVAL, "the CBET hooks exist for the default kernel (CBET_KERNEL_LDS_WINDOW) only");

    const cbet_derived &d = ctx->d;
    // the shipped kernel keeps two per-wave step counters in 16-bit halves of a register (WaveCounters)
    if (variant == CBET_KERNEL_LDS_WINDOW && d.nt >= 65536)
        return fail(CBET_EINVAL, "nt = %d steps per ray: the default kernel counts a bundle's steps in 16 bits (courant_mult too small)", d.nt);
    TraceArgs a{};
    a.nx = p->nx; a.ny = p->ny; a.nz = p->nz;
    a.xmin = p->xmin; a.ymin = p->ymin; a.zmin = p->zmin;
    a.dx = d.dx; a.dy = d.dy; a.dz = d.dz; a.dt = d.dt;
    a.inv_dx = 1 / d.dx; a.inv_dy = 1 / d.dy; a.inv_dz = 1 / d.dz;      // launch_ray_XZ.cu:276-278
    a.fx_hi = p->nx - 3.0; a.fy_hi = p->ny - 3.0; a.fz_hi = p->nz - 3.0;
    a.bounds = ctx->bounds;                                              // :352-354, see context_create
    a.tol_x = 0.5001 * d.dx; a.tol_y = 0.5001 * d.dy; a.tol_z = 0.5001 * d.dz;  // :164-176
    a.xconst = xconst; a.yconst = yconst; a.zconst = zconst;
    a.nt = d.nt; a.absorption = p->absorption;
    a.rpz = p->rays_per_zone; a.zones = d.zones_spanned; a.nrays_x = d.nrays_x;
    a.z_launch = kFocal - d.dz / 2;                                       // :97
    a.uray_mult = d.uray_mult; a.omega = d.omega; a.ncrit = d.ncrit;
    a.xlaunch = ctx->xlaunch; a.ylaunch = ctx->ylaunch;
    a.live = ctx->live; a.nlive = ctx->nlive;
    a.beam_lo = beam_lo; a.nbeams_local = beam_hi - beam_lo;
    a.bundles_per_beam = (ctx->nlive + kWave - 1) / kWave;
    a.total_bundles = (long)a.nbeams_local * a.bundles_per_beam;
    {   // this launch's share: a contiguous, near-equal part of the list (cbet_params.shard_index / shard_count)
        const long K = p->shard_count > 1 ? p->shard_count : 1, r = p->shard_count > 1 ? p->shard_index : 0;
        a.first_item = (r * a.total_bundles) / K;
        a.item_count = ((r + 1) * a.total_bundles) / K - a.first_item;
    }
    a.ne3d = ne3d ? ne3d : ctx->ne3d;
    a.kap3d = kappa3d ? kappa3d : ctx->kap3d;
    a.beam_norm = beam_norm; a.bbeam_norm = bbeam_norm; a.pow_r = pow_r; a.phase_r = phase_r;
    a.edep = edep;
    a.sYh = p->edep_zpitch > 0 ? p->edep_zpitch : p->nz + 2;
    a.sXh = (p->ny + 2) * a.sYh;
    if (p->edep_zpitch > 0 && (p->per_beam_grids || cbet_hooks))
        return fail(CBET_EINVAL, "edep_zpitch applies to the plain path's single deposit grid (no per-beam grids, no CBET hooks)");
    a.grid_stride = (p->per_beam_grids || hooks.quantity != 0) ? d.edep_size : 0;  // field passes are always beam-resolved
    // beam-resolved arrays may hold only the grids of beams [grid_beam0, grid_beam0 + grid_beams)
    const int gb_n = p->grid_beams > 0 ? p->grid_beams : p->nbeams, gb_0 = p->grid_beams > 0 ? p->grid_beam0 : 0;
    if ((a.grid_stride != 0 || hooks.gain) && (beam_lo < gb_0 || beam_hi > gb_0 + gb_n))
        return fail(CBET_EINVAL, "beams [%d,%d) are not all inside the beam-resolved arrays' range [%d,%d)", beam_lo, beam_hi,
                    gb_0, gb_0 + gb_n);
    a.grid_beam0 = gb_0;
    a.comp_stride = (long)gb_n * d.edep_size;
    a.counters = ctx->counters;
    a.stats = p->window_stats != 0;
    a.gain = hooks.gain; a.hsize = d.edep_size; a.quantity = hooks.quantity;
    a.max_exponent = hooks.max_exponent; a.beam_gain = hooks.beam_gain;
    DeviceGuard guard;
    CBET_HIP(hipSetDevice(ctx->gpu));
    if (variant == CBET_KERNEL_LDS_WINDOW) {
        // the shipped kernel gathers one 32-byte record per node; built here (unless still valid) because only
        // the launch knows both the tables -- possibly the caller's -- and the gradient constants
        if (int rc = step_records(ctx, p, ne3d, kappa3d, xconst, yconst, zconst, stream, false)) return rc;
        a.steprec = ctx->steprec;
    }
    CBET_HIP(launch_trace(a, variant, p->force_wide_index != 0, (hipStream_t)stream));
    return CBET_OK;
}

int cbet_trace_nodes(int b, unsigned nindices, const double *ne3d, const double *kappa3d,
                     double *edep, const double *bbeam_norm, const double *beam_norm,
                     const double *pow_r, const double *phase_r, double xconst, double yconst,
                     double zconst, const cbet_params *p, cbet_context *ctx, void *stream)
{
    return trace_impl(b, nindices, ne3d, kappa3d, edep, bbeam_norm, beam_norm, pow_r, phase_r, xconst, yconst,
                      zconst, p, ctx, stream, CbetHooks{});
}

// One lazily created workspace per device for callers that pass ctx == NULL (the reference's
// launch site has nothing to pass).  Recreated when the geometry changes.
static std::mutex g_ctx_mu;
static std::map<int, cbet_context *> g_default_ctx;

static int default_context(const cbet_params *p, cbet_context **out)
{
    int dev = 0;
    CBET_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_default_ctx.find(dev);
    if (it != g_default_ctx.end()) {
        if (check_geometry(it->second, p) == CBET_OK) {
            *out = it->second;
            return CBET_OK;
        }
        cbet_context_destroy(it->second);
        g_default_ctx.erase(it);
    }
    cbet_context *ctx = nullptr;
    if (int rc = cbet_context_create(&ctx, p, dev)) return rc;
    g_default_ctx[dev] = ctx;
    *out = ctx;
    return CBET_OK;
}

int cbet_launch_ray_XYZ(int b, unsigned nindices, double *te_data_g, double *r_data_g,
                        double *ne_data_g, double *edep, double *bbeam_norm, double *beam_norm,
                        double *pow_r, double *phase_r, double xconst, double yconst,
                        double zconst, const cbet_params *p, cbet_context *ctx, void *stream)
{
    if (int rc = validate(p)) return rc;
    if (!ctx) {
        if (int rc = default_context(p, &ctx)) return rc;
    }
    if (int rc = cbet_tabulate_plasma(ctx, p, te_data_g, r_data_g, ne_data_g, stream)) return rc;
    return cbet_trace_nodes(b, nindices, nullptr, nullptr, edep, bbeam_norm, beam_norm, pow_r, phase_r,
                            xconst, yconst, zconst, p, ctx, stream);
}

int cbet_edep_average_device(const double *edep, double *edepavg, int nx, int ny, int nz, void *stream)
{
    if (!edep || !edepavg || nx < 1 || ny < 1 || nz < 1) return fail(CBET_EINVAL, "cbet_edep_average_device: bad array");
    if ((long)(nx + 2) * (ny + 2) * (nz + 2) >= (1L << 31)) return fail(CBET_EINVAL, "cbet_edep_average_device: grid too large");
    CBET_HIP(launch_edep_average(edep, edepavg, nx, ny, nz, (hipStream_t)stream));
    return CBET_OK;
}

// ---- sparse exchange of the slab-owned CBET loop (cbet_grid_kernels.hip) -----------------------------------------
int cbet_pack_segments(const double *src, long beam_stride, int hy, int hz, const int *segments, long nseg, double *out,
                       void *stream)
{
    if (nseg < 0 || hy < 1 || hz < 1 || beam_stride < 0) return fail(CBET_EINVAL, "cbet_pack_segments: bad shape");
    if (nseg > 0 && (!src || !segments || !out)) return fail(CBET_EINVAL, "cbet_pack_segments: NULL pointer");
    if (nseg * 8 >= (1L << 31) * 256L) return fail(CBET_EINVAL, "cbet_pack_segments: too many segments for one launch");
    CBET_HIP(launch_pack_segments(src, beam_stride, hy, hz, segments, nseg, out, (hipStream_t)stream));
    return CBET_OK;
}

int cbet_unpack_segments(double *dst, long beam_stride, int hy, int hz, const int *segments, long nseg, const double *in,
                         void *stream)
{
    if (nseg < 0 || hy < 1 || hz < 1 || beam_stride < 0) return fail(CBET_EINVAL, "cbet_unpack_segments: bad shape");
    if (nseg > 0 && (!dst || !segments || !in)) return fail(CBET_EINVAL, "cbet_unpack_segments: NULL pointer");
    if (nseg * 8 >= (1L << 31) * 256L) return fail(CBET_EINVAL, "cbet_unpack_segments: too many segments for one launch");
    CBET_HIP(launch_unpack_segments(dst, beam_stride, hy, hz, segments, nseg, in, (hipStream_t)stream));
    return CBET_OK;
}

// ---- CBET stage (parity unpinned -- see the header) ---------------------------
static int validate_gain(const cbet_params *p, const cbet_gain_params *g)
{
    if (!g) return fail(CBET_EINVAL, "gain params is NULL");
    if (p->nbeams > CBET_MAX_CBET_BEAMS) return fail(CBET_EINVAL, "the CBET stage supports at most %d beams", CBET_MAX_CBET_BEAMS);
    if (!(g->max_exponent > 0.0 && g->max_exponent <= 1.0)) return fail(CBET_EINVAL, "max_exponent must be in (0, 1]");
    if (!(g->relax > 0.0 && g->relax <= 1.0)) return fail(CBET_EINVAL, "relax must be in (0, 1]");
    if (!(g->iaw > 0.0) || !(g->z_ion > 0.0) || !(g->te_ev > 0.0) || !(g->ti_ev >= 0.0) || !(g->mi_over_me > 0.0))
        return fail(CBET_EINVAL, "bad plasma constants in gain params");
    if (!(g->mach_r1 > g->mach_r0)) return fail(CBET_EINVAL, "mach_r1 must exceed mach_r0");
    if (g->direction_passes < 1) return fail(CBET_EINVAL, "direction_passes must be >= 1");
    return CBET_OK;
}

int cbet_gain_params_default(cbet_gain_params *g)
{
    if (!g) return fail(CBET_EINVAL, "gain params is NULL");
    std::memset(g, 0, sizeof *g);
    g->z_ion = 3.1;           // def.cuh:100
    g->te_ev = 2.0e3;         // def.cuh:104
    g->ti_ev = 1.0e3;         // def.cuh:106
    g->mi_over_me = 10230.0;  // def.cuh:101-102
    g->iaw = 0.2;             // def.cuh:107
    g->mach_r0 = 0.04; g->mach_0 = 0.4;   // def.cuh:114 names an undefined `machnum`; a radial ramp stands in
    g->mach_r1 = 0.13; g->mach_1 = 2.4;
    g->max_exponent = 1.0;
    g->relax = 0.5;           // plain fixed-point iteration (1.0) oscillates with 60 overlapping beams (scripts/cbet_converge.py)
    g->tolerance = 1e-4;
    g->max_passes = 40;
    g->direction_passes = 1;  // ray paths do not depend on the gain: the direction field of the gain-free first pass is kept
    return CBET_OK;
}

int cbet_gain_constants(const cbet_params *p, const cbet_gain_params *g, double *constant1, double *cs,
                        double *gain_const)
{
    if (int rc = validate(p)) return rc;
    if (int rc = validate_gain(p, g)) return rc;
    cbet_derived d;
    derive_core(p, &d);
    const double estat = 4.80320427e-10;            // def.cuh:98
    const double kb = 1.3806485279e-16;             // def.cuh:108
    const double te_k = g->te_ev * 11604.5052;      // def.cuh:103
    const double ti_k = g->ti_ev * 11604.5052;      // def.cuh:105
    const double mi_kg = g->mi_over_me * kMe;       // def.cuh:102
    const double c1 = (std::pow(estat, 2)) / (4 * (1.0e3 * kMe) * kC * d.omega * kb * te_k * (1 + 3 * ti_k / (g->z_ion * te_k)));  // def.cuh:111
    const double sound = 1e2 * std::sqrt(kEc * (g->z_ion * g->te_ev + 3.0 * g->ti_ev) / mi_kg);                                   // def.cuh:113
    if (constant1) *constant1 = c1;
    if (cs) *cs = sound;
    if (gain_const) *gain_const = c1 * (8.0 * M_PI * 1.0e7 / kC);  // |E|^2 = 8 pi 1e7 I / c
    return CBET_OK;
}

int cbet_trace_cbet(int b, unsigned nindices, const double *ne3d, const double *kappa3d,
                    const double *gain, int quantity, double *out, double *beam_gain,
                    const double *bbeam_norm, const double *beam_norm, const double *pow_r,
                    const double *phase_r, double xconst, double yconst, double zconst,
                    const cbet_params *p, const cbet_gain_params *g, cbet_context *ctx, void *stream)
{
    if (int rc = validate(p)) return rc;
    if (int rc = validate_gain(p, g)) return rc;
    if (quantity != CBET_DEPOSIT_ENERGY && quantity != CBET_DEPOSIT_FIELDS && quantity != CBET_DEPOSIT_FIELD_ENERGY)
        return fail(CBET_EINVAL, "quantity must be CBET_DEPOSIT_ENERGY (0), CBET_DEPOSIT_FIELDS (1) or CBET_DEPOSIT_FIELD_ENERGY (2)");
    CbetHooks h;
    h.gain = gain; h.quantity = quantity; h.beam_gain = beam_gain; h.max_exponent = g->max_exponent;
    return trace_impl(b, nindices, ne3d, kappa3d, out, bbeam_norm, beam_norm, pow_r, phase_r, xconst, yconst, zconst,
                      p, ctx, stream, h);
}

int cbet_gain_field(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                    const cbet_params *p, const cbet_gain_params *g, cbet_context *ctx, void *stream)
{
    return cbet_gain_field_slab(fields, ne3d, gain, scratch, change, 0, p ? p->nx + 2 : 0, p, g, ctx, stream);
}

static int gain_field_impl(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                           int hx_lo, int hx_hi, bool packed, const cbet_params *p, const cbet_gain_params *g,
                           cbet_context *ctx, void *stream, bool consume = false);

int cbet_gain_field_slab(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                         int hx_lo, int hx_hi, const cbet_params *p, const cbet_gain_params *g,
                         cbet_context *ctx, void *stream)
{
    return gain_field_impl(fields, ne3d, gain, scratch, change, hx_lo, hx_hi, false, p, g, ctx, stream);
}

int cbet_gain_field_packed(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                           int hx_lo, int hx_hi, const cbet_params *p, const cbet_gain_params *g,
                           cbet_context *ctx, void *stream)
{
    return gain_field_impl(fields, ne3d, gain, scratch, change, hx_lo, hx_hi, true, p, g, ctx, stream);
}

size_t cbet_cbet_slab_workspace_bytes_parts(const cbet_params *p, int own_beams, int own_planes, size_t staging_doubles)
{
    if (!p || validate(p) != CBET_OK || own_beams < 0 || own_beams > p->nbeams || own_planes < 0 || own_planes > p->nx + 2) return 0;
    const size_t plane = (size_t)(p->ny + 2) * (p->nz + 2), hsize = (size_t)(p->nx + 2) * plane, nb = (size_t)p->nbeams;
    // own beams over the whole grid: 4 field components + gain; all beams over the own slab: 4 components + gain
    // (the pair-once gain kernel keeps its sums in LDS: no scratch array since round 3; the dense exchange sends from and
    // receives into these arrays: no staging since round 4)
    return (5 * (size_t)own_beams * hsize + 5 * nb * (size_t)own_planes * plane + staging_doubles + 2 + CBET_MAX_CBET_BEAMS) * sizeof(double);
}

size_t cbet_cbet_slab_workspace_bytes(const cbet_params *p, int world_size, int rank)
{
    if (!p || validate(p) != CBET_OK || world_size < 1 || rank < 0 || rank >= world_size) return 0;
    // contiguous near-equal parts, as tracer._parts
    const size_t nb = (size_t)p->nbeams;
    const size_t own_beams = ((size_t)(rank + 1) * nb) / world_size - ((size_t)rank * nb) / world_size;
    const size_t own_planes = ((size_t)(rank + 1) * (p->nx + 2)) / world_size - ((size_t)rank * (p->nx + 2)) / world_size;
    return cbet_cbet_slab_workspace_bytes_parts(p, (int)own_beams, (int)own_planes, 0);
}

static int gain_field_impl(double *fields, const double *ne3d, double *gain, double *scratch, double *change,
                           int hx_lo, int hx_hi, bool packed, const cbet_params *p, const cbet_gain_params *g,
                           cbet_context *ctx, void *stream, bool consume)
{
    if (!ctx) return fail(CBET_EINVAL, "NULL context");
    if (int rc = validate(p)) return rc;
    if (int rc = check_geometry(ctx, p)) return rc;
    if (int rc = validate_gain(p, g)) return rc;
    if (!fields || !gain) return fail(CBET_EINVAL, "NULL device pointer");
    if (hx_lo < 0 || hx_hi > p->nx + 2 || hx_hi < hx_lo) return fail(CBET_EINVAL, "slab [%d,%d) outside the haloed grid [0,%d)", hx_lo, hx_hi, p->nx + 2);
    double cs = 0, gc = 0;
    if (int rc = cbet_gain_constants(p, g, nullptr, &cs, &gc)) return rc;
    const cbet_derived &d = ctx->d;
    GainArgs a{};
    a.nx = p->nx; a.ny = p->ny; a.nz = p->nz; a.nbeams = p->nbeams;
    a.xmin = p->xmin; a.ymin = p->ymin; a.zmin = p->zmin;
    a.dx = d.dx; a.dy = d.dy; a.dz = d.dz; a.dt = d.dt;
    a.ncrit = d.ncrit; a.k0 = d.omega / kC;
    a.cs = cs; a.gain_const = gc; a.iaw = g->iaw;
    a.mach_r0 = g->mach_r0; a.mach_0 = g->mach_0; a.mach_r1 = g->mach_r1; a.mach_1 = g->mach_1;
    a.relax = g->relax;
    a.fields = fields; a.ne3d = ne3d ? ne3d : ctx->ne3d; a.gain = gain; a.scratch = scratch; a.change = change;
    a.hx_lo = hx_lo; a.hx_hi = hx_hi;
    a.consume = (consume && scratch) ? 1 : 0;
    a.frozen = g->directions_frozen ? 1 : 0;
    const long plane = (long)(p->ny + 2) * (p->nz + 2);
    a.store0 = packed ? (long)hx_lo * plane : 0;
    a.bstride = packed ? (long)(hx_hi - hx_lo) * plane : d.edep_size;
    if (hx_hi == hx_lo) return CBET_OK;   // an empty slab (more ranks than planes)
    DeviceGuard guard;
    CBET_HIP(hipSetDevice(ctx->gpu));
    CBET_HIP(launch_gain_field(a, (hipStream_t)stream));
    return CBET_OK;
}

size_t cbet_cbet_workspace_bytes(const cbet_params *p)
{
    if (!p || validate(p) != CBET_OK) return 0;
    const size_t hsize = (size_t)(p->nx + 2) * (p->ny + 2) * (p->nz + 2);
    return (5 * (size_t)p->nbeams * hsize + 2 + CBET_MAX_CBET_BEAMS) * sizeof(double);   // 4 field components + gain
}

int cbet_cbet_solve(double *te_data_g, double *r_data_g, double *ne_data_g, double *edep,
                    double *bbeam_norm, double *beam_norm, double *pow_r, double *phase_r,
                    const cbet_params *p, const cbet_gain_params *g, void *workspace,
                    cbet_context *ctx, void *stream, cbet_cbet_report *report)
{
    if (int rc = validate(p)) return rc;
    if (int rc = validate_gain(p, g)) return rc;
    if (p->shard_count > 1) return fail(CBET_EINVAL, "cbet_cbet_solve runs on one device; the sharded loop is tracer.cbet_solve");
    if (g->max_passes < 1) return fail(CBET_EINVAL, "max_passes must be >= 1");
    if (!edep || !beam_norm || !pow_r || !phase_r) return fail(CBET_EINVAL, "NULL device pointer");
    if (!ctx) {
        if (int rc = default_context(p, &ctx)) return rc;
    }
    if (int rc = check_geometry(ctx, p)) return rc;
    const cbet_derived &d = ctx->d;
    hipStream_t s = (hipStream_t)stream;
    DeviceGuard guard;
    CBET_HIP(hipSetDevice(ctx->gpu));

    const size_t hsize = (size_t)d.edep_size, nb = (size_t)p->nbeams;
    const size_t bytes = cbet_cbet_workspace_bytes(p);
    double *ws = (double *)workspace;
    bool own = false;
    if (!ws) {
        hipError_t e = hipMalloc((void **)&ws, bytes);
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? CBET_ENOMEM : CBET_EHIP, "hipMalloc(cbet workspace, %zu bytes): %s", bytes, hipGetErrorString(e));
        own = true;
    }
    double *fields = ws, *gain = ws + 4 * nb * hsize, *change = gain + nb * hsize, *beam_gain = change + 2;
    double *const pair_once = gain;   // any non-NULL pointer selects the pair-once gain kernel; it is not dereferenced
    int rc = CBET_OK;
    cbet_counters c0{}, c1{};
    cbet_cbet_report rep{};
    auto body = [&]() -> int {
        if (int r = cbet_tabulate_plasma(ctx, p, te_data_g, r_data_g, ne_data_g, stream)) return r;
        if (int r = cbet_context_counters(ctx, stream, &c0, 0)) return r;
        CBET_HIP(hipMemsetAsync(gain, 0, nb * hsize * sizeof(double), s));
        cbet_params pf = *p;            // field passes: every beam (always beam-resolved)
        pf.beam_lo = 0; pf.beam_hi = p->nbeams;
        cbet_params pd = *p;            // deposition pass: the caller's grid layout, every beam
        pd.beam_lo = 0; pd.beam_hi = p->nbeams;
        // the fields are cleared once; every gain update hands them back zeroed (GainArgs.consume)
        CBET_HIP(hipMemsetAsync(fields, 0, 4 * nb * hsize * sizeof(double), s));
        cbet_gain_params gg = *g;
        for (int pass = 0; pass < g->max_passes; ++pass) {
            // the first direction_passes passes deposit all four fields and build k; later ones the energy field only
            const bool full = pass < g->direction_passes;
            if (full && pass > 0)   // a second direction-building pass accumulates into cleared direction entries
                CBET_HIP(hipMemsetAsync(fields + nb * hsize, 0, 3 * nb * hsize * sizeof(double), s));
            {
                CbetHooks h;
                h.gain = pass == 0 ? nullptr : gain; h.quantity = full ? CBET_DEPOSIT_FIELDS : CBET_DEPOSIT_FIELD_ENERGY;
                h.max_exponent = g->max_exponent;
                if (int r = trace_impl(0, (unsigned)d.nindices, nullptr, nullptr, fields, bbeam_norm, beam_norm, pow_r, phase_r,
                                       d.xconst, d.yconst, d.zconst, &pf, ctx, stream, h))
                    return r;
            }
            CBET_HIP(hipMemsetAsync(change, 0, 2 * sizeof(double), s));
            gg.directions_frozen = full ? 0 : 1;
            if (int r = gain_field_impl(fields, nullptr, gain, pair_once, change, 0, p->nx + 2, false, p, &gg, ctx, stream, true)) return r;
            double hc[2];
            CBET_HIP(hipMemcpyAsync(hc, change, sizeof hc, hipMemcpyDeviceToHost, s));
            CBET_HIP(hipStreamSynchronize(s));
            rep.passes = pass + 1;
            rep.change = hc[1] > 0.0 ? hc[0] / hc[1] : 0.0;
            if (rep.change < g->tolerance) { rep.converged = 1; break; }
        }
        CBET_HIP(hipMemsetAsync(beam_gain, 0, CBET_MAX_CBET_BEAMS * sizeof(double), s));
        if (int r = cbet_context_counters(ctx, stream, &c1, 0)) return r;
        rep.ray_steps = c1.ray_steps - c0.ray_steps;
        CbetHooks h;
        h.gain = gain; h.quantity = 0; h.beam_gain = beam_gain; h.max_exponent = g->max_exponent;
        if (int r = trace_impl(0, (unsigned)d.nindices, nullptr, nullptr, edep, bbeam_norm, beam_norm, pow_r, phase_r,
                               d.xconst, d.yconst, d.zconst, &pd, ctx, stream, h))
            return r;
        CBET_HIP(hipMemcpyAsync(rep.beam_gain, beam_gain, nb * sizeof(double), hipMemcpyDeviceToHost, s));
        cbet_counters c2{};
        if (int r = cbet_context_counters(ctx, stream, &c2, 0)) return r;   // synchronises
        rep.ray_steps_final = c2.ray_steps - c1.ray_steps;
        rep.ray_steps += rep.ray_steps_final;
        double net = 0.0, mag = 0.0;
        for (size_t bb = 0; bb < nb; ++bb) { net += rep.beam_gain[bb]; mag += std::fabs(rep.beam_gain[bb]); }
        rep.imbalance = mag > 0.0 ? std::fabs(net) / mag : 0.0;
        return CBET_OK;
    };
    rc = body();
    if (own) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(ws);
    }
    if (rc == CBET_OK && report) *report = rep;
    return rc;
}

}  // extern "C"
