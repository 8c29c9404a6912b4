"""A short randomised GPU-vs-oracle sweep (tools/fuzz_parity.py; longer sweeps are run by hand):
random meshes (height fields, partial / subdivided bunnies, triangle soups, nested spheres), sources,
time windows, samples per face, shading normals; every occlusion back-end must accept exactly the
oracle's samples."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.gpu
def test_random_scenes_match_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "16", "7"],
                         capture_output=True, text=True, timeout=900)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
    assert out.returncode == 0 and "mismatches 0 / 16" in tail, out.stdout[-3000:] + out.stderr[-1000:]
