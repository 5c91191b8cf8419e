# SPDX-License-Identifier: Apache-2.0
"""The endpoint coders against the reference's pack_color_endpoints / unpack_color_endpoints.

oracle/harness/compare_endpoint_coders.cpp links the reference's objects with the sequential CPU build of
astc-encoder_amd/csrc/wave_color.h (quad-lane LDR coders) and wave_color_hdr.h (HDR sub-modes side by side, driven by
width / spare-bit tables) and feeds both the same random and adversarial endpoint pairs: every requested format,
every colour quant level, bytes + format + decoded endpoints must be identical.  It also reports which HDR sub-mode
won, so that a run that never reached some sub-mode or escape layout fails here instead of passing silently.
Needs the reference objects, so it runs in the dev container only."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "compare_endpoint_coders")


@pytest.fixture(scope="module")
def harness(built):
    if not os.path.isdir("/root/reference/Source"):
        pytest.skip("reference sources not on this machine")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "coders"])
    return HARNESS


@pytest.mark.parametrize("seed", [1, 2])
def test_coders_match_reference(harness, seed):
    r = subprocess.run([harness, "20000", str(seed)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert re.search(r"OK \(0 mismatches in \d+ cases\)", r.stdout), r.stdout[-500:]
    # every produced format (incl. the base + offset ones, 5 / 9 / 13) and every HDR sub-mode + escape was exercised
    produced = dict((int(a), int(b)) for a, b in re.findall(r" (\d+):(\d+)", r.stdout.split("\n")[0]))
    assert all(produced[f] > 0 for f in range(16) if f != 1), produced      # (format 1, luminance delta, is never produced)
    fits = re.search(r"RGB\+offset((?: \d+)+); direct RGB((?: \d+)+); alpha((?: \d+)+)", r.stdout)
    for group, n in zip(fits.groups(), (6, 9, 4)):
        counts = [int(v) for v in group.split()]
        assert len(counts) == n and all(v > 0 for v in counts), fits.group(0)
