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

        def today():
            shutil.copytree(pkg, os.path.join(tmp, "cbet_raytracing_3d_amd"), ignore=shutil.ignore_patterns("lib", "__pycache__", "data"))
            shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))

        today()
        if os.path.exists(patch_path):
            # A patch names the commit it was cut against ("# Base: <sha>").  The kernel moves on; when the patch no longer
            # applies to today's tree the variant is built from `git archive <base>` of that commit instead, so that every
            # recorded experiment stays rebuildable as scripts/variants/README.md promises (with today's compiler flags).
            clean = subprocess.run(["patch", "-p1", "-s", "--dry-run", "-F0", "-i", patch_path], cwd=tmp, capture_output=True).returncode == 0
            if not clean:
                base = []
                for line in open(patch_path).read(2000).splitlines():       # "# Base: <sha>" or "... Base commit: <sha>."
                    if line.startswith("#") and "Base" in line:
                        words = line.replace(".", " ").replace(":", " ").split()
                        at = words.index(next(x for x in words if x.startswith("Base")))
                        base += [w for w in words[at + 1:] if len(w) >= 7 and all(c in "0123456789abcdef" for c in w)][:1]
                if not base:
                    raise SystemExit("%s does not apply to today's tree and names no '# Base: <commit>'" % patch_path)
                print("patch does not apply to HEAD: building from its base commit", base[0])
                for d in ("cbet_raytracing_3d_amd", "include"):
                    shutil.rmtree(os.path.join(tmp, d))
                tar = subprocess.run(["git", "archive", base[0], "cbet_raytracing_3d_amd", "include"], cwd=ROOT, check=True, capture_output=True).stdout
                subprocess.run(["tar", "-x", "-C", tmp], input=tar, check=True)
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
