"""scripts/variants/: every experiment that was measured and not shipped is kept as flags + a patch against the tree it was
cut from.  A patch that no longer applies is a result that can no longer be reproduced: each one must apply cleanly to the
sources it names (dry run; nothing is built here)."""
import glob
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

VDIR = os.path.join(ROOT, "scripts", "variants")
PATCHES = sorted(glob.glob(os.path.join(VDIR, "*.patch")))


def test_every_variant_has_flags_and_a_readme():
    assert os.path.exists(os.path.join(VDIR, "README.md"))
    names = {os.path.splitext(os.path.basename(p))[0] for p in PATCHES}
    for n in names:
        assert os.path.exists(os.path.join(VDIR, n + ".flags")), n
    for f in glob.glob(os.path.join(VDIR, "*.flags")):
        for tok in open(f).read().split():
            if tok.startswith("patch="):
                assert os.path.exists(os.path.join(VDIR, tok[len("patch="):])), (f, tok)


@pytest.mark.parametrize("patch", PATCHES, ids=[os.path.basename(p) for p in PATCHES])
def test_patch_applies_to_its_base(patch, tmp_path):
    """Against the commit the patch names in its header (`Base: <sha>` / `Base commit: <sha>`), not against today's tree:
    the product moves on, the record must stay reproducible from history."""
    head = open(patch).read(2000)
    base = None
    for line in head.splitlines():
        if line.startswith("#") and "Base" in line:
            words = line.replace(".", " ").replace(":", " ").split()
            base = next((w for w in words[words.index(next(x for x in words if x.startswith("Base"))) + 1:] if len(w) >= 7 and all(c in "0123456789abcdef" for c in w)), None)
    assert base, "the patch header must name its base commit"
    if shutil.which("git") is None or subprocess.run(["git", "cat-file", "-e", base + "^{commit}"], cwd=ROOT).returncode != 0:
        pytest.skip("history not available")
    arch = subprocess.run(["git", "archive", base, "cbet_raytracing_3d_amd", "include"], cwd=ROOT, capture_output=True, check=True).stdout
    subprocess.run(["tar", "x"], input=arch, cwd=tmp_path, check=True)
    run = subprocess.run(["patch", "-p1", "--dry-run", "-s", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
