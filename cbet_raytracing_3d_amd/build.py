"""Builds libcbet_mi355x.so (HIP kernels for gfx950 + the C ABI) in-tree with hipcc.

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container; the resulting
.so travels to the GPU box with the source snapshot.  `python -m cbet_raytracing_3d_amd.build`.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcbet_mi355x.so")
SOURCES = ["cbet_kernels.hip", "cbet_trace_window.hip", "cbet_grid_kernels.hip", "cbet_abi.cpp", "cbet_host.cpp", "cbet_output.cpp"]
HEADERS = [os.path.join(CSRC, "cbet_device.h"), os.path.join(CSRC, "cbet_relocate.h"), os.path.join(CSRC, "cbet_trace_common.h"), os.path.join(ROOT, "include", "cbet_mi355x.h"),
           os.path.join(ROOT, "include", "cbet_omega_beams.h")]

# -ffp-contract=off: a ray's fp64 arithmetic must be the reference's operation sequence (no fused
# multiply-add), see cbet_kernels.hip.  No -ffast-math: fp64 div/sqrt stay correctly rounded.
# -structurizecfg-skip-uniform-regions: the trace kernel's step loop branches on wave-uniform conditions only (scalar masks);
# structurised like divergent control flow, every such branch gets flow blocks whose phis cost the common path ~40 scalar
# copies per step (two per loop-carried value).  Uniform regions are left as the source wrote them.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wextra", "-Wno-unused-result", "-mllvm", "-structurizecfg-skip-uniform-regions=true"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm to build the gfx950 library)")
    return exe


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__),
                                                                os.path.join(ROOT, "tools", "cbet_gpu.cpp"),
                                                                os.path.join(ROOT, "tools", "cbet_reference_shaped.cpp"),
                                                                os.path.join(ROOT, "include", "cbet_reference_api.hpp")]
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + list(extra_flags) + [
        "-I", os.path.join(ROOT, "include"), "-I", CSRC,
        "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    run = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    sys.stderr.write(run.stderr)
    if run.returncode != 0 and "structurizecfg-skip-uniform-regions" in run.stderr:
        # a compiler without the (hidden) LLVM option: the library is the same without it, the trace kernel ~6 % slower
        print("build: this hipcc does not know -structurizecfg-skip-uniform-regions; building without it", flush=True)
        k = cmd.index("-structurizecfg-skip-uniform-regions=true")
        subprocess.check_call(cmd[:k - 1] + cmd[k + 1:])      # (the option and the -mllvm in front of it)
    elif run.returncode != 0:
        raise subprocess.CalledProcessError(run.returncode, cmd)
    build_cli(verbose)
    build_ref_shaped(verbose)
    return LIB_PATH


CLI_PATH = os.path.join(PKG, "lib", "cbet-gpu")


def build_cli(verbose=False):
    """tools/cbet_gpu.cpp -> lib/cbet-gpu: the main.cu-shaped driver, linked against the C ABI only."""
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "cbet_gpu.cpp"), "-o", CLI_PATH,
           "-L", LIB_DIR, "-lcbet_mi355x", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return CLI_PATH


REF_SHAPED_PATH = os.path.join(PKG, "lib", "cbet-ref-shaped")


def build_ref_shaped(verbose=False):
    """tools/cbet_reference_shaped.cpp -> lib/cbet-ref-shaped: rayTracing()'s call sequence through the C++
    overloads of include/cbet_reference_api.hpp (host-only code, compiled by hipcc for the HIP runtime API)."""
    cmd = [hipcc(), "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "cbet_reference_shaped.cpp"), "-o", REF_SHAPED_PATH,
           "-L", LIB_DIR, "-lcbet_mi355x", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return REF_SHAPED_PATH


def build_variant(name, extra_flags=(), verbose=False):
    """An experiment build of the same library, build_alt/libcbet_<name>.so (git-ignored; bench it with
    CBET_LIB_PATH, scripts/alt_sweep.sh).  Never shipped."""
    out_dir = os.path.join(ROOT, "build_alt")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libcbet_%s.so" % name)
    cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", out] + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-lrccl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":   # --variant NAME [flags...]
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
