# SPDX-License-Identifier: Apache-2.0
"""Wide bit-exactness sweep (tools/gpu_sweep.py): every 2D footprint x presets x image classes x
profiles against the reference library on the GPU box's host threads."""
import os
import subprocess
import sys
import oracle_libs as O  # (path set up by conftest.py)

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sweep_all_footprints_presets_profiles(product, A):
    if not os.path.exists(O.LIB_REF_AVX2):
        pytest.skip("oracle/_ref not present")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_sweep.py"), "120"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("sweep:") and " 0 mismatching cases" in last, out.stdout[-3000:]
