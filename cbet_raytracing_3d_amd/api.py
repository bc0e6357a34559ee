"""ctypes binding of the C ABI in include/cbet_mi355x.h.

This is plumbing over libcbet_mi355x.so (hand-written HIP for gfx950): every function here calls
straight into the library and raises CbetError when the library reports a failure.  There is no
CPU fallback -- if the shared library is missing, importing this module raises.

Names mirror the reference's interface for the path (file:line into /root/reference):
    safeGPUAlloc / moveToAndFromGPU   multi_gpu.cuh:6-7
    launch_ray_XYZ                    launch_ray_XZ.cu:117-121 (launch site main.cu:171-174)
    ray_tracing                       rayTracing(), main.cu:96-232
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# CBET_LIB_PATH: point at an alternative build of the same library (profiling experiments only)
LIB_PATH = os.environ.get("CBET_LIB_PATH") or os.path.join(_PKG, "lib", "libcbet_mi355x.so")
DATA_DIR = os.path.join(_PKG, "data")

OK, EINVAL, EHIP, ENOMEM, ENODEVICE, ECOMM = 0, -1, -2, -3, -4, -5
NPHASE = 2001
KERNEL_DEFAULT, KERNEL_GLOBAL_ATOMICS, KERNEL_LDS_COMBINE, KERNEL_LDS_WINDOW = 0, 1, 2, 3


class CbetError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cbet error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    """cbet_params (def.cuh:33-131 as run-time fields)."""
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("xmin", C.c_double), ("xmax", C.c_double),
        ("ymin", C.c_double), ("ymax", C.c_double),
        ("zmin", C.c_double), ("zmax", C.c_double),
        ("nbeams", C.c_int), ("rays_per_zone", C.c_int),
        ("courant_mult", C.c_double),
        ("absorption", C.c_int), ("nprofile", C.c_int),
        ("max_threads", C.c_int), ("threads_per_block", C.c_int),
        ("ngpus", C.c_int),
        ("beam_lo", C.c_int), ("beam_hi", C.c_int),
        ("shard_index", C.c_int), ("shard_count", C.c_int),
        ("kernel_variant", C.c_int), ("force_wide_index", C.c_int),
        ("per_beam_grids", C.c_int), ("patch_order", C.c_int),
        ("grid_beam0", C.c_int), ("grid_beams", C.c_int),
        ("rim_merge", C.c_int), ("edep_zpitch", C.c_int),
        ("window_stats", C.c_int),
    ]

    def copy(self, **overrides):
        q = Params()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(Params))
        for k, v in overrides.items():
            setattr(q, k, v)
        return q


class Derived(C.Structure):
    """cbet_derived (def.cuh / main.cu:156-161 derived constants)."""
    _fields_ = [
        ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double), ("dt", C.c_double),
        ("nt", C.c_int), ("zones_spanned", C.c_int),
        ("nrays_x", C.c_int), ("nrays_y", C.c_int), ("nrays", C.c_int),
        ("omega", C.c_double), ("ncrit", C.c_double), ("uray_mult", C.c_double),
        ("xconst", C.c_double), ("yconst", C.c_double), ("zconst", C.c_double),
        ("threads_per_beam", C.c_long), ("nindices", C.c_int), ("grid_y", C.c_int),
        ("edep_size", C.c_long), ("ntraced_ids", C.c_long), ("nlive_rays", C.c_long),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("ray_steps", C.c_ulonglong), ("rays_traced", C.c_ulonglong),
        ("global_atomics", C.c_ulonglong), ("lds_evictions", C.c_ulonglong),
        ("wave_steps", C.c_ulonglong), ("wave_steps_miss", C.c_ulonglong),
        ("wave_steps_wide", C.c_ulonglong), ("slabs_retired", C.c_ulonglong),
    ]


MAX_CBET_BEAMS = 64
DEPOSIT_ENERGY, DEPOSIT_FIELDS, DEPOSIT_FIELD_ENERGY = 0, 1, 2   # cbet_trace_cbet's `quantity`


class GainParams(C.Structure):
    """cbet_gain_params -- the CBET stage (parity unpinned: no reference counterpart)."""
    _fields_ = [
        ("z_ion", C.c_double), ("te_ev", C.c_double), ("ti_ev", C.c_double), ("mi_over_me", C.c_double),
        ("iaw", C.c_double),
        ("mach_r0", C.c_double), ("mach_0", C.c_double), ("mach_r1", C.c_double), ("mach_1", C.c_double),
        ("max_exponent", C.c_double), ("relax", C.c_double), ("tolerance", C.c_double),
        ("max_passes", C.c_int), ("direction_passes", C.c_int), ("directions_frozen", C.c_int), ("reserved_", C.c_int),
    ]


class CbetReport(C.Structure):
    _fields_ = [
        ("passes", C.c_int), ("converged", C.c_int), ("change", C.c_double), ("imbalance", C.c_double),
        ("beam_gain", C.c_double * MAX_CBET_BEAMS),
        ("ray_steps", C.c_ulonglong), ("ray_steps_final", C.c_ulonglong),
    ]


# Every symbol include/cbet_mi355x.h declares; tests check the library exports all of them.
EXPORTS = [
    "cbet_last_error", "cbet_version", "cbet_params_default", "cbet_derive",
    "cbet_live_ray_list", "cbet_omega60_beam_norm", "cbet_host_power_table", "cbet_host_beam_trig", "cbet_read_profile",
    "cbet_safeGPUAlloc", "cbet_moveToAndFromGPU", "cbet_gpuFree",
    "cbet_context_create", "cbet_context_destroy", "cbet_context_counters", "cbet_context_tables", "cbet_context_set_launch_list",
    "cbet_launch_ray_XYZ", "cbet_tabulate_plasma", "cbet_trace_nodes", "cbet_prepare_step_records", "cbet_ray_tracing",
    "cbet_write_text", "cbet_edep_average", "cbet_edep_average_device", "cbet_node_coordinates", "cbet_write_npy",
    "cbet_debug_bounds_violations",
    "cbet_gain_params_default", "cbet_gain_constants", "cbet_trace_cbet", "cbet_gain_field",
    "cbet_cbet_workspace_bytes", "cbet_cbet_solve", "cbet_gain_field_slab", "cbet_gain_field_packed",
    "cbet_cbet_slab_workspace_bytes", "cbet_cbet_slab_workspace_bytes_parts", "cbet_pack_segments", "cbet_unpack_segments",
]

_lib = None


def lib():
    """Loads libcbet_mi355x.so; raises (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process.  torch ships its own libamdhip64 / libhsa-runtime64 / librccl;
    # if libcbet_mi355x.so were loaded first it would pull in /opt/rocm's copies and the process would
    # hold two runtimes (the second to touch the device then fails, and streams / events of one mean
    # nothing to the other).  Importing torch first makes the loader resolve this library's HIP and
    # RCCL dependencies to the copies torch has already mapped (checked with scripts/which_hip.py),
    # so torch.cuda streams, events and tensors are valid arguments to every call below.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m cbet_raytracing_3d_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.cbet_last_error.restype = C.c_char_p
    L.cbet_version.restype = C.c_char_p
    L.cbet_params_default.argtypes = [C.POINTER(Params), C.c_int]
    L.cbet_derive.argtypes = [C.POINTER(Params), C.POINTER(Derived)]
    L.cbet_live_ray_list.argtypes = [C.POINTER(Params), ip, C.c_long, C.POINTER(C.c_long)]
    L.cbet_omega60_beam_norm.restype = dp
    L.cbet_host_power_table.argtypes = [dp, dp]
    L.cbet_host_beam_trig.argtypes = [dp, C.c_int, dp]
    L.cbet_read_profile.argtypes = [C.c_char_p, C.c_int, dp, dp]
    L.cbet_safeGPUAlloc.argtypes = [C.POINTER(vp), C.c_size_t, C.c_int]
    L.cbet_moveToAndFromGPU.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.cbet_gpuFree.argtypes = [vp, C.c_int]
    L.cbet_context_create.argtypes = [C.POINTER(vp), C.POINTER(Params), C.c_int]
    L.cbet_context_destroy.argtypes = [vp]
    L.cbet_context_counters.argtypes = [vp, vp, C.POINTER(Counters), C.c_int]
    L.cbet_context_set_launch_list.argtypes = [vp, ip, C.c_long]
    L.cbet_context_tables.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.cbet_launch_ray_XYZ.argtypes = [C.c_int, C.c_uint, vp, vp, vp, vp, vp, vp, vp, vp,
                                      C.c_double, C.c_double, C.c_double, C.POINTER(Params), vp, vp]
    L.cbet_tabulate_plasma.argtypes = [vp, C.POINTER(Params), vp, vp, vp, vp]
    L.cbet_prepare_step_records.argtypes = [vp, C.POINTER(Params), vp, vp, C.c_double, C.c_double, C.c_double, vp]
    L.cbet_trace_nodes.argtypes = [C.c_int, C.c_uint, vp, vp, vp, vp, vp, vp, vp,
                                   C.c_double, C.c_double, C.c_double, C.POINTER(Params), vp, vp]
    L.cbet_ray_tracing.argtypes = [dp, dp, dp, dp, C.POINTER(Params), dp, ip, C.c_int, dp,
                                   C.POINTER(Counters)]
    L.cbet_debug_bounds_violations.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, vp]
    L.cbet_write_text.argtypes = [dp, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.cbet_write_text.restype = C.c_longlong
    L.cbet_edep_average.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int]
    L.cbet_edep_average_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.cbet_node_coordinates.argtypes = [C.POINTER(Params), dp, dp, dp]
    L.cbet_write_npy.argtypes = [dp, C.c_int, C.POINTER(C.c_long), C.c_char_p]
    L.cbet_write_npy.restype = C.c_longlong
    L.cbet_gain_params_default.argtypes = [C.POINTER(GainParams)]
    L.cbet_gain_constants.argtypes = [C.POINTER(Params), C.POINTER(GainParams), dp, dp, dp]
    L.cbet_trace_cbet.argtypes = [C.c_int, C.c_uint, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp,
                                  C.c_double, C.c_double, C.c_double, C.POINTER(Params), C.POINTER(GainParams), vp, vp]
    L.cbet_gain_field.argtypes = [vp, vp, vp, vp, vp, C.POINTER(Params), C.POINTER(GainParams), vp, vp]
    L.cbet_gain_field_slab.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(GainParams), vp, vp]
    L.cbet_gain_field_packed.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(GainParams), vp, vp]
    L.cbet_pack_segments.argtypes = [vp, C.c_long, C.c_int, C.c_int, vp, C.c_long, vp, vp]
    L.cbet_unpack_segments.argtypes = [vp, C.c_long, C.c_int, C.c_int, vp, C.c_long, vp, vp]
    L.cbet_cbet_slab_workspace_bytes.argtypes = [C.POINTER(Params), C.c_int, C.c_int]
    L.cbet_cbet_slab_workspace_bytes.restype = C.c_size_t
    L.cbet_cbet_slab_workspace_bytes_parts.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_size_t]
    L.cbet_cbet_slab_workspace_bytes_parts.restype = C.c_size_t
    L.cbet_cbet_workspace_bytes.argtypes = [C.POINTER(Params)]
    L.cbet_cbet_workspace_bytes.restype = C.c_size_t
    L.cbet_cbet_solve.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(Params), C.POINTER(GainParams), vp, vp, vp,
                                  C.POINTER(CbetReport)]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here = the library is older than the header
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise CbetError(rc, lib().cbet_last_error().decode("utf-8", "replace"))


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _addr(x):
    """Device/host address of a torch tensor, numpy array, ctypes pointer or int (None -> NULL)."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if isinstance(x, C.c_void_p):
        return x.value
    raise TypeError("cannot take the address of %r" % type(x))


# ---- configuration -----------------------------------------------------------------------------
def version():
    return lib().cbet_version().decode()


def default_params(n=100, **overrides):
    p = Params()
    _check(lib().cbet_params_default(C.byref(p), n))
    for k, v in overrides.items():
        setattr(p, k, v)
    return p


def derive(p):
    d = Derived()
    _check(lib().cbet_derive(C.byref(p), C.byref(d)))
    return d


def live_ray_list(p):
    """Launch list in kernel order: 64 consecutive entries form one bundle, -1 marks a hole
    (see cbet_live_ray_list in the header)."""
    n = C.c_long()
    _check(lib().cbet_live_ray_list(C.byref(p), None, 0, C.byref(n)))
    out = np.zeros(n.value, dtype=np.int32)
    _check(lib().cbet_live_ray_list(C.byref(p), out.ctypes.data_as(C.POINTER(C.c_int)), n.value,
                                    C.byref(n)))
    return out


def shard_items(p, nbeams_local, shard_index, shard_count):
    """Host-side statement of the kernel's work split: the (beam_local, thread-ray id) pairs that
    shard `shard_index` of `shard_count` traces (a contiguous near-equal part of the beam-major bundle list)."""
    live = live_ray_list(p)
    bpb = (len(live) + 63) // 64
    beams, ids = [], []
    total = nbeams_local * bpb
    K, r = max(1, shard_count), (shard_index if shard_count > 1 else 0)
    for g in range((r * total) // K, ((r + 1) * total) // K):
        b, k = g // bpb, g % bpb
        chunk = live[64 * k: 64 * k + 64]
        chunk = chunk[chunk >= 0]
        beams.append(np.full(len(chunk), b, dtype=np.int32))
        ids.append(chunk)
    if not ids:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    return np.concatenate(beams), np.concatenate(ids)


def omega60_beam_norm():
    ptr = lib().cbet_omega60_beam_norm()
    return np.ctypeslib.as_array(ptr, shape=(60, 3)).copy()


def host_power_table():
    phase, powr = np.zeros(NPHASE), np.zeros(NPHASE)
    _check(lib().cbet_host_power_table(_dptr(phase), _dptr(powr)))
    return phase, powr


def host_beam_trig(beam_norm):
    bn = np.ascontiguousarray(beam_norm, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((bn.shape[0], 4))
    _check(lib().cbet_host_beam_trig(_dptr(bn), bn.shape[0], _dptr(out)))
    return out


def read_profile(path, nprofile=443):
    r, v = np.zeros(nprofile), np.zeros(nprofile)
    _check(lib().cbet_read_profile(os.fsencode(path), nprofile, _dptr(r), _dptr(v)))
    return r, v


def load_s83177(nprofile=443):
    """(r, ne, te) of the s83177 shot at 1.5 ns, first `nprofile` rows (main.cu:249-260:
    the Te file is read first, then the ne file, whose radii overwrite the first's)."""
    _, te = read_profile(os.path.join(DATA_DIR, "s83177_te.txt"), nprofile)
    r, ne = read_profile(os.path.join(DATA_DIR, "s83177_ne.txt"), nprofile)
    return r, ne, te


# ---- multi_gpu.cuh helpers ---------------------------------------------------------------------
def safeGPUAlloc(size, gpu):
    """multi_gpu.cpp:3-28.  Returns the device address (int)."""
    out = C.c_void_p()
    _check(lib().cbet_safeGPUAlloc(C.byref(out), size, gpu))
    return out.value


def moveToAndFromGPU(dst, src, size, gpu):
    """multi_gpu.cpp:44-59.  dst/src: addresses, numpy arrays or torch tensors."""
    _check(lib().cbet_moveToAndFromGPU(_addr(dst), _addr(src), size, gpu))


def gpuFree(ptr, gpu):
    _check(lib().cbet_gpuFree(_addr(ptr), gpu))


# ---- workspace ---------------------------------------------------------------------------------
class Context:
    """cbet_context: per-device workspace (node tables, live-ray list, counters)."""

    def __init__(self, params, gpu=0):
        self._h = C.c_void_p()
        self.gpu = gpu
        _check(lib().cbet_context_create(C.byref(self._h), C.byref(params), gpu))

    @property
    def handle(self):
        return self._h

    def counters(self, stream=None, reset=False):
        c = Counters()
        _check(lib().cbet_context_counters(self._h, _addr(stream), C.byref(c), 1 if reset else 0))
        return c

    def set_launch_list(self, slots):
        """Regroup the bundles: `slots` holds the context's live rays, each once, 64 entries per bundle, -1 = idle lane."""
        arr = np.ascontiguousarray(slots, dtype=np.int32)
        _check(lib().cbet_context_set_launch_list(self._h, arr.ctypes.data_as(C.POINTER(C.c_int)), arr.size))

    def tables(self):
        a, b = C.c_void_p(), C.c_void_p()
        _check(lib().cbet_context_tables(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self._h:
            lib().cbet_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- the hot path ------------------------------------------------------------------------------
def launch_ray_XYZ(b, nindices, te_data_g, r_data_g, ne_data_g, edep, bbeam_norm, beam_norm, pow_r,
                   phase_r, xconst, yconst, zconst, params, ctx=None, stream=None):
    """launch_ray_XZ.cu:117-121 argument for argument, plus params / workspace / stream.
    Pointer arguments are device addresses (ints) or torch tensors."""
    _check(lib().cbet_launch_ray_XYZ(
        b, nindices, _addr(te_data_g), _addr(r_data_g), _addr(ne_data_g), _addr(edep),
        _addr(bbeam_norm), _addr(beam_norm), _addr(pow_r), _addr(phase_r), xconst, yconst, zconst,
        C.byref(params), ctx.handle if ctx is not None else None, _addr(stream)))


def tabulate_plasma(ctx, params, te_data_g, r_data_g, ne_data_g, stream=None):
    _check(lib().cbet_tabulate_plasma(ctx.handle, C.byref(params), _addr(te_data_g),
                                      _addr(r_data_g), _addr(ne_data_g), _addr(stream)))


def prepare_step_records(ctx, params, ne3d, kappa3d, xconst, yconst, zconst, stream=None):
    """Build the default kernel's per-node step records now instead of inside the next launch (see the header)."""
    _check(lib().cbet_prepare_step_records(ctx.handle, C.byref(params), _addr(ne3d), _addr(kappa3d), xconst, yconst,
                                           zconst, _addr(stream)))


def trace_nodes(b, nindices, ne3d, kappa3d, edep, bbeam_norm, beam_norm, pow_r, phase_r, xconst,
                yconst, zconst, params, ctx, stream=None):
    _check(lib().cbet_trace_nodes(
        b, nindices, _addr(ne3d), _addr(kappa3d), _addr(edep), _addr(bbeam_norm), _addr(beam_norm),
        _addr(pow_r), _addr(phase_r), xconst, yconst, zconst, C.byref(params), ctx.handle,
        _addr(stream)))


# ---- CBET stage (parity unpinned) -----------------------------------------------
def default_gain_params(**overrides):
    g = GainParams()
    _check(lib().cbet_gain_params_default(C.byref(g)))
    for k, v in overrides.items():
        setattr(g, k, v)
    return g


def gain_constants(params, gain_params):
    """(constant1, cs, gain_const) of def.cuh:111, 113."""
    out = [C.c_double() for _ in range(3)]
    _check(lib().cbet_gain_constants(C.byref(params), C.byref(gain_params), *[C.byref(o) for o in out]))
    return tuple(o.value for o in out)


def trace_cbet(b, nindices, ne3d, kappa3d, gain, quantity, out, beam_gain, bbeam_norm, beam_norm, pow_r,
               phase_r, xconst, yconst, zconst, params, gain_params, ctx, stream=None):
    _check(lib().cbet_trace_cbet(
        b, nindices, _addr(ne3d), _addr(kappa3d), _addr(gain), quantity, _addr(out), _addr(beam_gain),
        _addr(bbeam_norm), _addr(beam_norm), _addr(pow_r), _addr(phase_r), xconst, yconst, zconst,
        C.byref(params), C.byref(gain_params), ctx.handle, _addr(stream)))


def gain_field(fields, ne3d, gain, scratch, change, params, gain_params, ctx, stream=None):
    _check(lib().cbet_gain_field(_addr(fields), _addr(ne3d), _addr(gain), _addr(scratch), _addr(change),
                                 C.byref(params), C.byref(gain_params), ctx.handle, _addr(stream)))


def gain_field_packed(fields, ne3d, gain, scratch, change, hx_lo, hx_hi, params, gain_params, ctx, stream=None):
    """cbet_gain_field_slab on slab-packed arrays (planes [hx_lo, hx_hi) of every beam only)."""
    _check(lib().cbet_gain_field_packed(_addr(fields), _addr(ne3d), _addr(gain), _addr(scratch), _addr(change), hx_lo, hx_hi,
                                        C.byref(params), C.byref(gain_params), ctx.handle, _addr(stream)))


def pack_segments(src, beam_stride, hy, hz, segments, nseg, out, stream=None):
    """cbet_pack_segments: gather the 64-byte z-runs listed in `segments` (device int32 [nseg][2]) of `src` into `out`."""
    _check(lib().cbet_pack_segments(_addr(src), beam_stride, hy, hz, _addr(segments), nseg, _addr(out), _addr(stream)))


def unpack_segments(dst, beam_stride, hy, hz, segments, nseg, buf, stream=None):
    """cbet_unpack_segments: scatter `buf` (nseg runs of 8 doubles) into the listed runs of `dst`."""
    _check(lib().cbet_unpack_segments(_addr(dst), beam_stride, hy, hz, _addr(segments), nseg, _addr(buf), _addr(stream)))


def cbet_slab_workspace_bytes(params, world_size, rank):
    return int(lib().cbet_cbet_slab_workspace_bytes(C.byref(params), world_size, rank))


def cbet_slab_workspace_bytes_parts(params, own_beams, own_planes, staging_doubles=0):
    return int(lib().cbet_cbet_slab_workspace_bytes_parts(C.byref(params), own_beams, own_planes, staging_doubles))


def gain_field_slab(fields, ne3d, gain, scratch, change, hx_lo, hx_hi, params, gain_params, ctx, stream=None):
    _check(lib().cbet_gain_field_slab(_addr(fields), _addr(ne3d), _addr(gain), _addr(scratch), _addr(change), hx_lo, hx_hi,
                                      C.byref(params), C.byref(gain_params), ctx.handle, _addr(stream)))


def cbet_workspace_bytes(params):
    return int(lib().cbet_cbet_workspace_bytes(C.byref(params)))


def cbet_solve(te_data_g, r_data_g, ne_data_g, edep, bbeam_norm, beam_norm, pow_r, phase_r, params,
               gain_params, workspace=None, ctx=None, stream=None):
    rep = CbetReport()
    _check(lib().cbet_cbet_solve(_addr(te_data_g), _addr(r_data_g), _addr(ne_data_g), _addr(edep),
                                 _addr(bbeam_norm), _addr(beam_norm), _addr(pow_r), _addr(phase_r),
                                 C.byref(params), C.byref(gain_params), _addr(workspace),
                                 ctx.handle if ctx is not None else None, _addr(stream), C.byref(rep)))
    return rep


def ray_tracing(te_profile, r_profile, ne_profile, edep, params, beam_norm=None, gpus=None, ngpu=1):
    """rayTracing() (main.cu:96-232): host profiles in, host `edep` (numpy, (nx+2,ny+2,nz+2))
    ADDED into.  Returns (timers{init,tracing,combining,total}, Counters)."""
    te = np.ascontiguousarray(te_profile, dtype=np.float64)
    r = np.ascontiguousarray(r_profile, dtype=np.float64)
    ne = np.ascontiguousarray(ne_profile, dtype=np.float64)
    if not (isinstance(edep, np.ndarray) and edep.dtype == np.float64 and edep.flags.c_contiguous):
        raise TypeError("edep must be a C-contiguous float64 numpy array")
    bn = None
    if beam_norm is not None:
        bn = np.ascontiguousarray(beam_norm, dtype=np.float64)
    garr = None
    if gpus is not None:
        garr = (C.c_int * len(gpus))(*gpus)
        ngpu = len(gpus)
    timers = np.zeros(4)
    cnt = Counters()
    _check(lib().cbet_ray_tracing(_dptr(te), _dptr(r), _dptr(ne), _dptr(edep), C.byref(params),
                                  _dptr(bn) if bn is not None else None, garr, ngpu, _dptr(timers),
                                  C.byref(cnt)))
    return dict(zip(("init", "tracing", "combining", "total"), timers.tolist())), cnt


# ---- output stage ------------------------------------------------------------------------------
def write_text(edep, path):
    """main.cu:6-22 `-D PRINT` rendering (truth_100's format) of a host (d0,d1,d2) array."""
    e = np.ascontiguousarray(edep, dtype=np.float64)
    n = lib().cbet_write_text(_dptr(e), e.shape[0], e.shape[1], e.shape[2],
                              os.fsencode(path) if path is not None else None)
    if n < 0:
        _check(int(n))
    return int(n)


def edep_average(edep):
    """main.cu:334-349: 27-point average of the haloed grid -> (nx, ny, nz)."""
    e = np.ascontiguousarray(edep, dtype=np.float64)
    nx, ny, nz = (d - 2 for d in e.shape)
    out = np.zeros((nx, ny, nz))
    _check(lib().cbet_edep_average(_dptr(e), _dptr(out), nx, ny, nz))
    return out


def node_coordinates(params):
    """main.cu:321-332: (x, y, z), each [nx][ny][nz]."""
    shape = (params.nx, params.ny, params.nz)
    x, y, z = np.empty(shape), np.empty(shape), np.empty(shape)
    _check(lib().cbet_node_coordinates(C.byref(params), _dptr(x), _dptr(y), _dptr(z)))
    return x, y, z


def write_npy(array, path):
    """The library's .npy writer (stand-in for the reference's dead HDF5 output); returns bytes written."""
    a = np.ascontiguousarray(array, dtype=np.float64)
    shape = (C.c_long * a.ndim)(*a.shape)
    n = lib().cbet_write_npy(_dptr(a), a.ndim, shape, os.fsencode(path))
    if n < 0:
        _check(int(n))
    return int(n)


def edep_average_device(edep, out, nx, ny, nz, stream=None):
    """main.cu:334-349 on the device: edep (n+2)^3 and out n^3 are device tensors / addresses."""
    _check(lib().cbet_edep_average_device(_addr(edep), _addr(out), nx, ny, nz, _addr(stream)))


def debug_bounds_violations(reset=True, stream=None):
    """Bounds-audit builds only: out-of-range accesses the kernels attempted since the last reset."""
    n = C.c_ulonglong()
    _check(lib().cbet_debug_bounds_violations(C.byref(n), 1 if reset else 0, _addr(stream)))
    return int(n.value)
