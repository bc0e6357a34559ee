// cbet_host.cpp -- cbet_ray_tracing(): the MI355X counterpart of rayTracing()
// (/root/reference/main.cu:96-232).  Same phases -- host tables, per-device upload, one launch
// per device from its own host thread, combine, timers -- with two differences that are the point
// of the redesign:
//   * rays are split across devices as CONTIGUOUS, equal parts of the beam-major list of ray bundles
//     (cbet_params.shard_index / shard_count: device g of G traces bundles [T g / G, T (g+1) / G) of the T),
//     not as blocks of nbeams/nGPUs whole beams (launch_ray_XZ.cu:123), so 60 beams on 8 devices lose
//     nothing to integer division, the load is even to one bundle, and a device walks only its own ~7.5
//     beams' stretch of the record table;
//   * the per-device grids are summed on the devices by one RCCL reduce-scatter over xGMI into x-slabs
//     (half the traffic of an all-reduce), and every device then copies ITS slab to the host over its own
//     PCIe link, in parallel, where its host thread ADDS it into the caller's grid -- replacing the serial
//     whole-grid D2H copies and the host += loop (main.cu:178-210) and keeping rayTracing()'s "edep +="
//     contract.  Communicators are created once per device set and cached for the life of the process.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cbet_mi355x.h"

namespace cbet {
int fail(int code, const char *fmt, ...);
}

namespace {

double now_s()
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double>(clk::now().time_since_epoch()).count();
}

// RCCL communicators per ordered device set, created on first use and kept (ncclCommInitAll costs tens of
// milliseconds -- more than a whole 256^3 pass).
std::mutex g_comm_mu;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;

ncclResult_t communicators(const std::vector<int> &devs, std::vector<ncclComm_t> **out)
{
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto it = g_comms.find(devs);
    if (it == g_comms.end()) {
        std::vector<ncclComm_t> c(devs.size());
        const ncclResult_t r = ncclCommInitAll(c.data(), (int)devs.size(), devs.data());
        if (r != ncclSuccess) return r;
        it = g_comms.emplace(devs, std::move(c)).first;
    }
    *out = &it->second;
    return ncclSuccess;
}

struct DeviceJob {
    int gpu = 0;
    int rc = CBET_OK;
    std::string err;
    cbet_context *ctx = nullptr;
    double *d_beam_norm = nullptr, *d_bbeam_norm = nullptr, *d_pow_r = nullptr, *d_phase_r = nullptr;
    double *d_ne = nullptr, *d_te = nullptr, *d_r = nullptr, *d_edep = nullptr, *d_slab = nullptr;
    hipStream_t stream = nullptr;
    cbet_counters counters{};
};

void note(DeviceJob &j, int rc)
{
    if (rc != CBET_OK && j.rc == CBET_OK) {
        j.rc = rc;
        j.err = cbet_last_error();
    }
}

void release(DeviceJob &j)
{
    if (hipSetDevice(j.gpu) == hipSuccess) {
        if (j.stream) (void)hipStreamDestroy(j.stream);
        cbet_context_destroy(j.ctx);
        for (double *p : {j.d_beam_norm, j.d_bbeam_norm, j.d_pow_r, j.d_phase_r, j.d_ne, j.d_te, j.d_r, j.d_edep, j.d_slab})
            if (p) (void)hipFree(p);
    }
    (void)hipGetLastError();  // a device that never existed must not leave a sticky error behind
}

}  // namespace

namespace {
// rayTracing() leaves the process on whatever device it touched last (and ends with
// cudaDeviceReset, main.cu:217); here the caller's current device is restored instead.
struct RestoreDevice {
    int saved = -1;
    RestoreDevice() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~RestoreDevice() { if (saved >= 0) (void)hipSetDevice(saved); (void)hipGetLastError(); }
};
}  // namespace

extern "C" int cbet_ray_tracing(const double *te_profile, const double *r_profile,
                                const double *ne_profile, double *edep, const cbet_params *p,
                                const double *beam_norm, const int *gpus, int ngpu, double *timers,
                                cbet_counters *counters)
{
    RestoreDevice restore_device;
    if (!te_profile || !r_profile || !ne_profile || !edep || !p)
        return cbet::fail(CBET_EINVAL, "cbet_ray_tracing: NULL argument");
    if (ngpu < 1) return cbet::fail(CBET_EINVAL, "cbet_ray_tracing: ngpu < 1");
    cbet_derived d;
    if (int rc = cbet_derive(p, &d)) return rc;
    if (!beam_norm) {
        if (p->nbeams > 60) return cbet::fail(CBET_EINVAL, "nbeams > 60 needs an explicit beam_norm table");
        beam_norm = cbet_omega60_beam_norm();
    }
    const double t1 = now_s();

    // main.cu:102-110, 121-129: host tables
    std::vector<double> phase_r(CBET_NPHASE), pow_r(CBET_NPHASE), bbeam(4 * (size_t)p->nbeams);
    cbet_host_power_table(phase_r.data(), pow_r.data());
    cbet_host_beam_trig(beam_norm, p->nbeams, bbeam.data());

    // the device grids are padded to a whole number of x-planes per device, so that the reduce-scatter hands
    // every device an equal slab (the padding planes stay zero)
    // CBET_FORCE_RCCL=1: run the RCCL combine on one device too (a one-rank communicator: the reduce-scatter hands the
    // device its own grid) -- lets a one-GPU box execute the communicator cache, ncclReduceScatter and the slab
    // download that multi-device runs use (tests/test_gpu_rccl_smoke.py)
    const char *force_env = std::getenv("CBET_FORCE_RCCL");
    const bool use_rccl = ngpu > 1 || (force_env && force_env[0] == '1');
    const size_t plane = (size_t)(p->ny + 2) * (p->nz + 2);
    const size_t slab_planes = ((size_t)p->nx + 2 + ngpu - 1) / ngpu, slab_elems = slab_planes * plane;
    const size_t edep_bytes = sizeof(double) * slab_elems * ngpu;
    const size_t nr = (size_t)p->nprofile;
    std::vector<DeviceJob> jobs(ngpu);
    for (int i = 0; i < ngpu; ++i) jobs[i].gpu = gpus ? gpus[i] : i;
    {
        std::vector<int> sorted(ngpu);
        for (int i = 0; i < ngpu; ++i) sorted[i] = jobs[i].gpu;
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
            return cbet::fail(CBET_EINVAL, "cbet_ray_tracing: a device is listed twice (one rank per device)");
    }

    // main.cu:133-152: allocate + upload per device (plus the workspace, and a zeroed grid: the
    // reference accumulates into memory it never clears)
    auto setup = [&](DeviceJob &j) {
        const int g = j.gpu;
        note(j, cbet_safeGPUAlloc((void **)&j.d_beam_norm, sizeof(double) * 3 * p->nbeams, g));
        note(j, cbet_safeGPUAlloc((void **)&j.d_bbeam_norm, sizeof(double) * 4 * p->nbeams, g));
        note(j, cbet_safeGPUAlloc((void **)&j.d_pow_r, sizeof(double) * CBET_NPHASE, g));
        note(j, cbet_safeGPUAlloc((void **)&j.d_phase_r, sizeof(double) * CBET_NPHASE, g));
        note(j, cbet_safeGPUAlloc((void **)&j.d_ne, sizeof(double) * nr, g));
        note(j, cbet_safeGPUAlloc((void **)&j.d_te, sizeof(double) * nr, g));
        note(j, cbet_safeGPUAlloc((void **)&j.d_r, sizeof(double) * nr, g));
        note(j, cbet_safeGPUAlloc((void **)&j.d_edep, edep_bytes, g));
        if (use_rccl) note(j, cbet_safeGPUAlloc((void **)&j.d_slab, sizeof(double) * slab_elems, g));
        if (j.rc) return;
        note(j, cbet_moveToAndFromGPU(j.d_beam_norm, (void *)beam_norm, sizeof(double) * 3 * p->nbeams, g));
        note(j, cbet_moveToAndFromGPU(j.d_bbeam_norm, bbeam.data(), sizeof(double) * 4 * p->nbeams, g));
        note(j, cbet_moveToAndFromGPU(j.d_pow_r, pow_r.data(), sizeof(double) * CBET_NPHASE, g));
        note(j, cbet_moveToAndFromGPU(j.d_phase_r, phase_r.data(), sizeof(double) * CBET_NPHASE, g));
        note(j, cbet_moveToAndFromGPU(j.d_ne, (void *)ne_profile, sizeof(double) * nr, g));
        note(j, cbet_moveToAndFromGPU(j.d_te, (void *)te_profile, sizeof(double) * nr, g));
        note(j, cbet_moveToAndFromGPU(j.d_r, (void *)r_profile, sizeof(double) * nr, g));
        if (j.rc) return;
        note(j, cbet_context_create(&j.ctx, p, g));
        if (j.rc) return;
        if (hipSetDevice(g) != hipSuccess || hipStreamCreate(&j.stream) != hipSuccess ||
            hipMemsetAsync(j.d_edep, 0, edep_bytes, j.stream) != hipSuccess ||
            hipStreamSynchronize(j.stream) != hipSuccess) {
            j.rc = CBET_EHIP;
            j.err = "stream/memset setup failed";
        }
    };
    {
        std::vector<std::thread> th;
        for (auto &j : jobs) th.emplace_back(setup, std::ref(j));
        for (auto &t : th) t.join();
    }
    auto first_error = [&]() -> int {
        for (auto &j : jobs)
            if (j.rc) return cbet::fail(j.rc, "device %d: %s", j.gpu, j.err.c_str());
        return CBET_OK;
    };
    if (int rc = first_error()) {
        for (auto &j : jobs) release(j);
        return rc;
    }
    const double t2 = now_s();

    // main.cu:166-176: one host thread per device launches and waits
    auto run = [&](DeviceJob &j, int index) {
        cbet_params q = *p;
        q.beam_lo = 0;
        q.beam_hi = p->nbeams;
        q.shard_index = index;
        q.shard_count = ngpu;
        (void)hipSetDevice(j.gpu);
        note(j, cbet_launch_ray_XYZ(index, (unsigned)d.nindices, j.d_te, j.d_r, j.d_ne, j.d_edep,
                                    j.d_bbeam_norm, j.d_beam_norm, j.d_pow_r, j.d_phase_r, d.xconst,
                                    d.yconst, d.zconst, &q, j.ctx, j.stream));
        if (j.rc) return;
        note(j, cbet_context_counters(j.ctx, j.stream, &j.counters, 0));  // synchronises the stream
    };
    {
        std::vector<std::thread> th;
        for (int i = 0; i < ngpu; ++i) th.emplace_back(run, std::ref(jobs[i]), i);
        for (auto &t : th) t.join();
    }
    if (int rc = first_error()) {
        for (auto &j : jobs) release(j);
        return rc;
    }
    const double t3 = now_s();

    // Combine (replaces main.cu:178-210): RCCL reduce-scatter over xGMI into x-slabs, then every device's host
    // thread copies its slab down its own PCIe link and adds it into the caller's grid ("edep +=", main.cu:206).
    int rc = CBET_OK;
    if (use_rccl) {
        std::vector<int> devs(ngpu);
        for (int i = 0; i < ngpu; ++i) devs[i] = jobs[i].gpu;
        std::vector<ncclComm_t> *comms = nullptr;
        ncclResult_t nr_ = communicators(devs, &comms);
        if (nr_ != ncclSuccess) {
            rc = cbet::fail(CBET_ECOMM, "ncclCommInitAll: %s", ncclGetErrorString(nr_));
        } else {
            ncclGroupStart();
            for (int i = 0; i < ngpu && nr_ == ncclSuccess; ++i) {
                (void)hipSetDevice(jobs[i].gpu);
                nr_ = ncclReduceScatter(jobs[i].d_edep, jobs[i].d_slab, slab_elems, ncclDouble, ncclSum, (*comms)[i],
                                        jobs[i].stream);
            }
            const ncclResult_t ge = ncclGroupEnd();
            if (nr_ == ncclSuccess) nr_ = ge;
            if (nr_ != ncclSuccess) rc = cbet::fail(CBET_ECOMM, "ncclReduceScatter: %s", ncclGetErrorString(nr_));
        }
        if (rc != CBET_OK) {
            // part of the group may have been enqueued: nothing may still be running on a device when release()
            // frees its buffers below
            for (auto &j : jobs) {
                (void)hipSetDevice(j.gpu);
                (void)hipStreamSynchronize(j.stream);
            }
            (void)hipGetLastError();
        }
    }
    if (rc == CBET_OK) {
        auto fetch = [&](DeviceJob &j, int index) {
            // slab `index` = planes [index * slab_planes, ...) of the haloed grid, clipped to nx + 2
            const size_t first = (size_t)index * slab_planes;
            const size_t planes_here = first >= (size_t)p->nx + 2 ? 0 : std::min(slab_planes, (size_t)p->nx + 2 - first);
            if (planes_here == 0) return;
            const size_t n = planes_here * plane;
            std::vector<double> staging(n);
            (void)hipSetDevice(j.gpu);
            const double *src = use_rccl ? j.d_slab : j.d_edep;
            if (hipMemcpyAsync(staging.data(), src, n * sizeof(double), hipMemcpyDeviceToHost, j.stream) != hipSuccess ||
                hipStreamSynchronize(j.stream) != hipSuccess) {
                j.rc = CBET_EHIP;
                j.err = "slab copy to the host failed";
                (void)hipGetLastError();
                return;
            }
            double *dst = edep + first * plane;
            for (size_t i = 0; i < n; ++i) dst[i] += staging[i];   // main.cu:206, slabs are disjoint
        };
        std::vector<std::thread> th;
        for (int i = 0; i < ngpu; ++i) th.emplace_back(fetch, std::ref(jobs[i]), i);
        for (auto &t : th) t.join();
        rc = first_error();
    }
    if (counters) {
        std::memset(counters, 0, sizeof *counters);
        for (auto &j : jobs) {
            counters->ray_steps += j.counters.ray_steps;
            counters->rays_traced += j.counters.rays_traced;
            counters->global_atomics += j.counters.global_atomics;
            counters->lds_evictions += j.counters.lds_evictions;
            counters->wave_steps += j.counters.wave_steps;
            counters->wave_steps_miss += j.counters.wave_steps_miss;
            counters->wave_steps_wide += j.counters.wave_steps_wide;
            counters->slabs_retired += j.counters.slabs_retired;
        }
    }
    for (auto &j : jobs) release(j);
    const double t4 = now_s();
    if (timers) {  // main.cu:219-231: Init, Tracing, Combining, Total
        timers[0] = t2 - t1;
        timers[1] = t3 - t2;
        timers[2] = t4 - t3;
        timers[3] = t4 - t1;
    }
    return rc;
}
