"""ctypes front end of the CPU oracle (oracle/cbet_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  See oracle/cbet_oracle.h for what the
oracle restates (reference file:line per function) and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcbet_oracle.so")
NPHASE = 2001


class Config(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
        ("xmin", C.c_double), ("xmax", C.c_double),
        ("ymin", C.c_double), ("ymax", C.c_double),
        ("zmin", C.c_double), ("zmax", C.c_double),
        ("nbeams", C.c_int), ("rays_per_zone", C.c_int),
        ("courant_mult", C.c_double),
        ("absorption", C.c_int), ("nprofile", C.c_int),
        ("max_threads", C.c_int), ("threads_per_block", C.c_int),
    ]


class Derived(C.Structure):
    _fields_ = [
        ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double), ("dt", C.c_double),
        ("nt", C.c_int), ("zones_spanned", C.c_int), ("nrays_x", C.c_int), ("nrays_y", C.c_int),
        ("nrays", C.c_int),
        ("omega", C.c_double), ("ncrit", C.c_double), ("uray_mult", C.c_double),
        ("xconst", C.c_double), ("yconst", C.c_double), ("zconst", C.c_double),
        ("threads_per_beam", C.c_long), ("nindices", C.c_int), ("grid_y", C.c_int),
        ("edep_size", C.c_long),
    ]


class GainConfig(C.Structure):
    """CBET extension (parity unpinned) -- see cbet_oracle.h."""
    _fields_ = [
        ("z_ion", C.c_double), ("te_ev", C.c_double), ("ti_ev", C.c_double), ("mi_over_me", C.c_double),
        ("iaw", C.c_double),
        ("mach_r0", C.c_double), ("mach_0", C.c_double), ("mach_r1", C.c_double), ("mach_1", C.c_double),
        ("max_exponent", C.c_double),
    ]


def build(force=False):
    """Compile oracle/libcbet_oracle.so with gcc (no GPU, no reference sources involved)."""
    src = os.path.join(_HERE, "cbet_oracle.c")
    hdr = os.path.join(_HERE, "cbet_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libcbet_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    L.cbet_oracle_default_config.argtypes = [C.POINTER(Config), C.c_int]
    L.cbet_oracle_derive.argtypes = [C.POINTER(Config), C.POINTER(Derived)]
    L.cbet_oracle_span.argtypes = [C.c_double, C.c_double, C.c_uint, _dp]
    L.cbet_oracle_power_table.argtypes = [_dp, _dp]
    L.cbet_oracle_beam_trig.argtypes = [_dp, C.c_int, _dp]
    L.cbet_oracle_interp.argtypes = [_dp, _dp, C.c_double, C.c_int]
    L.cbet_oracle_interp.restype = C.c_double
    L.cbet_oracle_launch_point.argtypes = [C.POINTER(Config), _dp, C.c_int, C.c_int, _dp, _dp, _dp]
    L.cbet_oracle_launch_point.restype = C.c_int
    L.cbet_oracle_id_is_traced.argtypes = [C.POINTER(Config), C.c_int]
    L.cbet_oracle_id_is_traced.restype = C.c_int
    L.cbet_oracle_trace.argtypes = [C.POINTER(Config), _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp,
                                    C.c_int, C.c_void_p]
    L.cbet_oracle_trace.restype = C.c_longlong
    L.cbet_oracle_trace_tables.argtypes = [C.POINTER(Config), _dp, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int]
    L.cbet_oracle_trace_tables.restype = C.c_longlong
    L.cbet_oracle_trace_list.argtypes = [C.POINTER(Config), _dp, _dp, _dp, _dp, C.c_long, _ip, _ip,
                                         _dp, C.c_int]
    L.cbet_oracle_trace_list.restype = C.c_longlong
    L.cbet_oracle_ray_path.argtypes = [C.POINTER(Config), _dp, _dp, _dp, _dp, C.c_int, C.c_int,
                                       C.c_int, _dp]
    L.cbet_oracle_ray_path.restype = C.c_int
    L.cbet_oracle_write_text.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.cbet_oracle_write_text.restype = C.c_longlong
    L.cbet_oracle_node_tables.argtypes = [C.POINTER(Config), _dp, _dp, _dp, _dp, C.c_void_p]
    L.cbet_oracle_gain_default.argtypes = [C.POINTER(GainConfig)]
    L.cbet_oracle_gain_constants.argtypes = [C.POINTER(Config), C.POINTER(GainConfig)] + [C.POINTER(C.c_double)] * 3
    L.cbet_oracle_trace_cbet.argtypes = [C.POINTER(Config), C.POINTER(GainConfig), _dp, _dp, _dp, C.c_void_p,
                                         C.c_int, C.c_int, _dp, C.c_void_p, C.c_int]
    L.cbet_oracle_trace_cbet.restype = C.c_longlong
    L.cbet_oracle_trace_cbet_list.argtypes = [C.POINTER(Config), C.POINTER(GainConfig), _dp, _dp, _dp, C.c_void_p,
                                              C.c_int, C.c_int, C.c_long, _ip, _ip, _dp, C.c_void_p, C.c_int]
    L.cbet_oracle_trace_cbet_list.restype = C.c_longlong
    L.cbet_oracle_phi.argtypes = [C.c_double]
    L.cbet_oracle_phi.restype = C.c_double
    L.cbet_oracle_gain_field.argtypes = [C.POINTER(Config), C.POINTER(GainConfig), _dp, _dp, C.c_double, _dp,
                                         _dp, C.c_int]
    _lib = L
    return L


def default_config(n, **overrides):
    cfg = Config()
    lib().cbet_oracle_default_config(C.byref(cfg), n)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def derive(cfg):
    d = Derived()
    lib().cbet_oracle_derive(C.byref(cfg), C.byref(d))
    return d


def power_table():
    phase = np.zeros(NPHASE)
    powr = np.zeros(NPHASE)
    lib().cbet_oracle_power_table(phase, powr)
    return phase, powr


def beam_trig(beam_norm):
    bn = np.ascontiguousarray(beam_norm, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((bn.shape[0], 4))
    lib().cbet_oracle_beam_trig(bn, bn.shape[0], out)
    return out


def interp(y, x, xp):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().cbet_oracle_interp(y, x, float(xp), len(x))


def launch_point(cfg, beam_norm, beam, raynum):
    phase, powr = power_table()
    out = np.zeros(4)
    live = lib().cbet_oracle_launch_point(C.byref(cfg), np.ascontiguousarray(beam_norm), beam,
                                          raynum, powr, phase, out)
    return bool(live), out


def grid_shape(cfg):
    return (cfg.nx + 2, cfg.ny + 2, cfg.nz + 2)


def trace(cfg, beam_norm, r, ne, te, beam_lo=0, beam_hi=None, nthreads=1, edep=None,
          want_per_beam=False):
    """Returns (edep[(nx+2),(ny+2),(nz+2)], ray_steps[, steps_per_beam])."""
    if beam_hi is None:
        beam_hi = cfg.nbeams
    if edep is None:
        edep = np.zeros(grid_shape(cfg))
    per_beam = np.zeros(cfg.nbeams, dtype=np.int64)
    steps = lib().cbet_oracle_trace(C.byref(cfg), np.ascontiguousarray(beam_norm, dtype=np.float64),
                                    np.ascontiguousarray(r), np.ascontiguousarray(ne),
                                    np.ascontiguousarray(te), beam_lo, beam_hi, edep, nthreads,
                                    per_beam.ctypes.data_as(C.c_void_p))
    if want_per_beam:
        return edep, int(steps), per_beam
    return edep, int(steps)


def trace_tables(cfg, beam_norm, ne3d, kap3d, beam_lo=0, beam_hi=None, nthreads=1):
    """Trace with caller-supplied node tables (3-D plasma).  Returns (edep, ray_steps)."""
    if beam_hi is None:
        beam_hi = cfg.nbeams
    edep = np.zeros(grid_shape(cfg))
    steps = lib().cbet_oracle_trace_tables(C.byref(cfg), np.ascontiguousarray(beam_norm, dtype=np.float64),
                                           np.ascontiguousarray(ne3d, dtype=np.float64),
                                           np.ascontiguousarray(kap3d, dtype=np.float64), beam_lo, beam_hi,
                                           edep, nthreads)
    return edep, int(steps)


def trace_list(cfg, beam_norm, r, ne, te, beams, raynums, nthreads=1, edep=None):
    if edep is None:
        edep = np.zeros(grid_shape(cfg))
    beams = np.ascontiguousarray(beams, dtype=np.int32)
    raynums = np.ascontiguousarray(raynums, dtype=np.int32)
    steps = lib().cbet_oracle_trace_list(C.byref(cfg), np.ascontiguousarray(beam_norm, dtype=np.float64),
                                         np.ascontiguousarray(r), np.ascontiguousarray(ne),
                                         np.ascontiguousarray(te), len(beams), beams, raynums, edep,
                                         nthreads)
    return edep, int(steps)


def ray_path(cfg, beam_norm, r, ne, te, beam, raynum, max_steps=None):
    if max_steps is None:
        max_steps = derive(cfg).nt
    path = np.zeros((max_steps, 8))
    n = lib().cbet_oracle_ray_path(C.byref(cfg), np.ascontiguousarray(beam_norm, dtype=np.float64),
                                   np.ascontiguousarray(r), np.ascontiguousarray(ne),
                                   np.ascontiguousarray(te), beam, raynum, max_steps, path)
    return path[:n]


def write_text(edep, path):
    e = np.ascontiguousarray(edep, dtype=np.float64)
    return int(lib().cbet_oracle_write_text(e, e.shape[0], e.shape[1], e.shape[2],
                                            os.fsencode(path)))


def node_tables(cfg, r, ne, te):
    ne3d = np.zeros((cfg.nx, cfg.ny, cfg.nz))
    kap = np.zeros((cfg.nx, cfg.ny, cfg.nz))
    lib().cbet_oracle_node_tables(C.byref(cfg), np.ascontiguousarray(r), np.ascontiguousarray(ne),
                                  np.ascontiguousarray(te), ne3d, kap.ctypes.data_as(C.c_void_p))
    return ne3d, kap


# ---- CBET extension (parity unpinned; checker of the HIP implementation of DESIGN.md section 9) ----
def gain_default(**overrides):
    g = GainConfig()
    lib().cbet_oracle_gain_default(C.byref(g))
    for k, v in overrides.items():
        setattr(g, k, v)
    return g


def gain_constants(cfg, g):
    """(constant1, cs, gain_const) of def.cuh:111,113."""
    out = [C.c_double() for _ in range(3)]
    lib().cbet_oracle_gain_constants(C.byref(cfg), C.byref(g), *[C.byref(o) for o in out])
    return tuple(o.value for o in out)


def phi(x):
    return lib().cbet_oracle_phi(float(x))


def trace_cbet(cfg, g, beam_norm, ne3d, kap3d, gain=None, quantity=0, per_beam=False, nthreads=1, items=None):
    """Returns (out, ray_steps, beam_gain[nbeams]).  items = (beams, raynums): trace that list only."""
    shape = ((cfg.nbeams,) if per_beam else ()) + grid_shape(cfg)
    out = np.zeros(shape)
    beam_gain = np.zeros(cfg.nbeams)
    gp = None
    if gain is not None:
        gain = np.ascontiguousarray(gain, dtype=np.float64)
        assert gain.size == cfg.nbeams * (cfg.nx + 2) * (cfg.ny + 2) * (cfg.nz + 2)
        gp = gain.ctypes.data_as(C.c_void_p)
    if items is not None:
        beams = np.ascontiguousarray(items[0], dtype=np.int32)
        raynums = np.ascontiguousarray(items[1], dtype=np.int32)
        steps = lib().cbet_oracle_trace_cbet_list(
            C.byref(cfg), C.byref(g), np.ascontiguousarray(beam_norm, dtype=np.float64),
            np.ascontiguousarray(ne3d, dtype=np.float64), np.ascontiguousarray(kap3d, dtype=np.float64), gp, quantity,
            1 if per_beam else 0, len(beams), beams, raynums, out, beam_gain.ctypes.data_as(C.c_void_p), nthreads)
        return out, int(steps), beam_gain
    steps = lib().cbet_oracle_trace_cbet(C.byref(cfg), C.byref(g), np.ascontiguousarray(beam_norm, dtype=np.float64),
                                         np.ascontiguousarray(ne3d, dtype=np.float64),
                                         np.ascontiguousarray(kap3d, dtype=np.float64), gp, quantity,
                                         1 if per_beam else 0, out, beam_gain.ctypes.data_as(C.c_void_p), nthreads)
    return out, int(steps), beam_gain


def gain_field(cfg, g, fields, ne3d, relax=1.0, gain=None, nthreads=1):
    """fields[4][nbeams][(n+2)^3] -> (gain[nbeams][(n+2)^3], (sum |new-old|, sum |new|))."""
    if gain is None:
        gain = np.zeros((cfg.nbeams,) + grid_shape(cfg))
    fields = np.ascontiguousarray(fields, dtype=np.float64)
    assert fields.size == 4 * cfg.nbeams * (cfg.nx + 2) * (cfg.ny + 2) * (cfg.nz + 2)
    change = np.zeros(2)
    lib().cbet_oracle_gain_field(C.byref(cfg), C.byref(g), fields, np.ascontiguousarray(ne3d, dtype=np.float64),
                                 float(relax), gain, change, nthreads)
    return gain, (float(change[0]), float(change[1]))
