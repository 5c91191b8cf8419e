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


ALL_SCRIPT = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import astcenc_amd as A, oracle_libs as O
lib = A.Library(O.LIB_EMU)
for bx, by in ((4, 4), (5, 4), (5, 5), (6, 5), (6, 6), (8, 5), (8, 6), (8, 8), (10, 5), (10, 6), (10, 8), (10, 10), (12, 10), (12, 12)):
    for q in (0.0, 10.0, 60.0, 98.0, 100.0):
        sys.stderr.write("case %%dx%%d q=%%g\n" %% (bx, by, q))
        lib.compress(A.synthetic_image(2 * bx, 2 * by), (bx, by), q)
    sys.stderr.write("case %%dx%%d hdr\n" %% (bx, by))
    lib.compress(A.synthetic_hdr_image(2 * bx, 2 * by, 1), (bx, by), 60.0, profile=A.PRF_HDR)
""" % (os.path.join(ROOT, "astc-encoder_amd", "python"), os.path.join(ROOT, "oracle"))


def test_candidate_records_of_a_batch_end_inside_the_allocation(built):
    """The batched first refinement step leaves one record per candidate but the first from LdsLayout::cstate on; a trial
    class laid out later may move cstate up (make_lds_layout): the last record of the largest batch must still end inside
    the block's allocation -- for every 2D footprint, five quality levels and the HDR profile."""
    env = dict(os.environ, ASTC_EMU_DUMP_LAYOUT="1")
    r = subprocess.run([sys.executable, "-c", ALL_SCRIPT], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    seen = 0
    for line in r.stderr.splitlines():
        m = re.search(r"lds layout: total (\d+) .* batch: cstate (\d+) stride (\d+) per batch (\d+) / (\d+)", line)
        if not m:
            continue
        total, cstate, stride, nb1, nb2 = (int(x) for x in m.groups())
        assert cstate + (max(nb1, nb2) - 1) * stride <= total, line
        seen += 1
    assert seen == 14 * 6, seen
