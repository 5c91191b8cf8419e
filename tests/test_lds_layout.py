# SPDX-License-Identifier: Apache-2.0
"""The LDS working set per block fixes the occupancy of the compression kernel: the hardware hands LDS out in units of
1280 bytes, 128 units per CU, and the register budget allows 16 workgroups per CU.  This guards
the footprints the BASELINE configs run at: 6x6 -medium must stay within 8 units (16 workgroups per CU), 8x8 -thorough
within 12 (10 workgroups per CU).  The layout comes from make_lds_layout() through the sequential build of the kernel
source (oracle/emu), which prints it on request; no GPU needed."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRANULE = 1280

SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import astcenc_amd as A, oracle_libs as O
lib = A.Library(O.LIB_EMU)
img = A.synthetic_image(24, 24)
bx, q, hdr = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
if hdr:
    import numpy as np
    lib.compress(A.synthetic_hdr_image(2 * bx, 2 * bx, 1), (bx, bx), q, profile=A.PRF_HDR)
else:
    lib.compress(img[:2 * bx, :2 * bx].copy(), (bx, bx), q)
""" % (os.path.join(ROOT, "astc-encoder_amd", "python"), os.path.join(ROOT, "oracle"))


def layout_total(block, quality, hdr=False):
    env = dict(os.environ, ASTC_EMU_DUMP_LAYOUT="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT, str(block), str(quality), str(int(hdr))], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"lds layout: total (\d+)", r.stderr)
    assert m, r.stderr[-2000:]
    return int(m.group(1))


@pytest.mark.parametrize("block,quality,hdr,units", [(6, 60.0, False, 8), (8, 98.0, False, 12), (6, 60.0, True, 8), (4, 60.0, False, 8)])
def test_working_set_stays_within_its_allocation_units(built, block, quality, hdr, units):
    total = layout_total(block, quality, hdr)
    assert total <= units * GRANULE, "%dx%d q=%g: %d B of LDS per block = %d units, more than %d: occupancy drops" % (
        block, block, quality, total, -(-total // GRANULE), units)
