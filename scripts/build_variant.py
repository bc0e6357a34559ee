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
    out_dir = os.path.join(ROOT, "build_alt")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(ROOT, "cbet_raytracing_3d_amd", "csrc"), os.path.join(tmp, "cbet_raytracing_3d_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
        if os.path.exists(patch_path):
            subprocess.check_call(["patch", "-p1", "-s", "-i", patch_path], cwd=tmp)
        csrc = os.path.join(tmp, "cbet_raytracing_3d_amd", "csrc")
        for suffix, extra in (("", []), ("_audit", ["-DCBET_DEBUG_BOUNDS"])):
            out = os.path.join(out_dir, "libcbet_%s%s.so" % (name, suffix))
            cmd = [B.hipcc()] + B.FLAGS + flags + extra + ["-I", os.path.join(tmp, "include"), "-I", csrc, "-o", out] + \
                  [os.path.join(csrc, s) for s in B.SOURCES] + ["-lrccl"]
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            print(out)


if __name__ == "__main__":
    main()
