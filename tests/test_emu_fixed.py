# SPDX-License-Identifier: Apache-2.0
"""The kernel source compiled for ONE context on the CPU (oracle/emu/Makefile `fixed`): what a fixed-context
build of the library and every run-time build (csrc/kernel_jit.cpp) are on the device -- LdsLayout, DeviceConfig and TableRoot
as compile-time constants, the ASTC_FIXED-only code paths (the unrolled mode scoring, for_texels_of_quarter, the literal
channel weights) compiled in -- as a sequential build, against the reference's bytes.  The records are the ones the plain
sequential build writes for the context (ASTC_EMU_DUMP_RECORDS: the text a run-time build is compiled with).

One footprint of at most 64 texels (the BASELINE headline context) and one above (10x8: the footprint whose run-time build
on the device needs the texel count kept out of the constants, wave_ctx.h)."""
import os
import subprocess

import numpy as np
import pytest

import images
import oracle_libs as O  # (path set up by conftest.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "oracle", "emu")


@pytest.mark.parametrize("block,quality,profile_name", [((6, 6), 60.0, "PRF_LDR"), ((10, 8), 10.0, "PRF_LDR_SRGB")])
def test_sequential_build_compiled_for_one_context(built, emu, ref, A, tmp_path, monkeypatch, block, quality, profile_name):
    profile = getattr(A, profile_name)
    name = "%dx%d_%d_%s" % (block[0], block[1], int(quality), profile_name.lower())
    records = str(tmp_path / (name + ".inc"))
    monkeypatch.setenv("ASTC_EMU_DUMP_RECORDS", records)
    img = images.noisy(66, 50, 21)
    plain = emu.compress(img, block, quality, profile=profile)
    monkeypatch.delenv("ASTC_EMU_DUMP_RECORDS")
    text = open(records).read()
    assert "constexpr LdsLayout kFixedLayout" in text and "constexpr TableRoot kFixedRoot" in text
    r = subprocess.run(["make", "-s", "fixed", "NAME=" + name, "RECORDS=" + records], cwd=EMU_DIR, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lib_path = os.path.join(EMU_DIR, "_build", "libastcenc_emu_fixed_%s.so" % name)
    try:
        fixed = A.Library(lib_path)
        want = ref.compress(img, block, quality, profile=profile)
        got = fixed.compress(img, block, quality, profile=profile)
        assert np.array_equal(want, plain)
        assert np.array_equal(want, got), int((want.reshape(-1, 16) != got.reshape(-1, 16)).any(axis=1).sum())
        # ... and like a fixed-context kernel build it turns every other context away
        err, cfg = fixed.config_init(profile, block[0], block[1], 1, A.PRE_THOROUGH, 0)
        assert err == 0
        err, ctx = fixed.context_alloc(cfg, 1)
        assert err != 0
    finally:
        os.remove(lib_path)
