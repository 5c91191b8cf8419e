# SPDX-License-Identifier: Apache-2.0
"""The table blob the kernels read (astc-encoder_amd/csrc/host_tables.cpp) against the reference's
block_size_descriptor and static tables, field by field (oracle/harness/compare_tables.cpp): block modes
in search order, decimation infos (bilinear 2D / simplex 3D), partition tables, coverage bitmaps, k-means
texels, quant / BISE / sin-cos tables.  Needs the reference objects, so it runs in the dev container only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "compare_tables")

CASES = [(4, 4, 1, 60), (6, 6, 1, 60), (6, 6, 1, 10), (8, 8, 1, 98), (5, 4, 1, 100), (10, 6, 1, 60), (12, 12, 1, 60),
         (3, 3, 3, 60), (4, 3, 3, 10), (4, 4, 4, 98), (5, 5, 4, 60), (6, 5, 5, 60), (6, 6, 6, 100)]


@pytest.fixture(scope="module")
def harness(built):
    if not os.path.isdir("/root/reference/Source"):
        pytest.skip("reference sources not on this machine")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "harness"])
    return HARNESS


@pytest.mark.parametrize("bx,by,bz,quality", CASES)
def test_tables_match_reference(harness, bx, by, bz, quality):
    r = subprocess.run([harness, str(bx), str(by), str(quality), str(bz)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "OK (0 mismatches)" in r.stdout
