#!/usr/bin/env python3
"""Build an experiment variant of the library from scripts/variants/<name>.flags (+ optional <name>.patch):
build_alt/libcbet_<name>.so and its bounds-audited twin build_alt/libcbet_<name>_audit.so.  See scripts/variants/README.md.

    python scripts/build_variant.py NAME [--new "-DFLAG ..."]     (--new writes NAME.flags first)
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cbet_raytracing_3d_amd import build as B   # noqa: E402

VDIR = os.path.join(ROOT, "scripts", "variants")


def main():
    name = sys.argv[1]
    if len(sys.argv) > 3 and sys.argv[2] == "--new":
        with open(os.path.join(VDIR, name + ".flags"), "w") as f:
            f.write(sys.argv[3].strip() + "\n")
    flags_path, patch_path = os.path.join(VDIR, name + ".flags"), os.path.join(VDIR, name + ".patch")
    if not os.path.exists(flags_path):
        raise SystemExit("no scripts/variants/%s.flags (create it, or pass --new \"flags\")" % name)
    flags = open(flags_path).read().split()
    for f in [f for f in flags if f.startswith("patch=")]:      # a variant may name another variant's patch
        patch_path = os.path.join(VDIR, f[len("patch="):])
        flags.remove(f)
    out_dir = os.path.join(ROOT, "build_alt")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        pkg = os.path.join(ROOT, "cbet_raytracing_3d_amd")
        shutil.copytree(pkg, os.path.join(tmp, "cbet_raytracing_3d_amd"), ignore=shutil.ignore_patterns("lib", "__pycache__", "data"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
        if os.path.exists(patch_path):
            subprocess.check_call(["patch", "-p1", "-s", "-i", patch_path], cwd=tmp)
            print("applied", patch_path, "(its Python parts -- api constants, tracer options -- are NOT installed: apply the patch to the tree "
                  "to drive a variant that needs them)")
        csrc = os.path.join(tmp, "cbet_raytracing_3d_amd", "csrc")
        sources = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))     # a patch may add files
        for suffix, extra in (("", []), ("_audit", ["-DCBET_DEBUG_BOUNDS"])):
            out = os.path.join(out_dir, "libcbet_%s%s.so" % (name, suffix))
            cmd = [B.hipcc()] + B.FLAGS + flags + extra + ["-I", os.path.join(tmp, "include"), "-I", csrc, "-o", out] + \
                  [os.path.join(csrc, s) for s in sources] + ["-lrccl"]
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            print(out)


if __name__ == "__main__":
    main()
