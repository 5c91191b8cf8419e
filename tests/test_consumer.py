# SPDX-License-Identifier: Apache-2.0
"""A real C++ consumer of the drop-in (VERDICT r01 "missing" 6): the reference's own API example,
Utils/Example/astc_api_example.cpp, compiled from where it lies by oracle/Makefile (`example`) against
include/astcenc.h and linked to libastcenc_amd.so (oracle/_ref/astc_api_example_amd), next to the same
translation unit linked to the reference library (oracle/_ref/astc_api_example_ref).

Here (no GPU): the binary's undefined astcenc_* symbols are all exported by the product library, i.e. the
header and the library agree on the ABI a C++ caller sees.  On the GPU box: both binaries load a PNG,
compress it 6x6 -medium, decompress it and write a PNG; the two outputs must be identical images."""
import os
import subprocess

import numpy as np
import pytest

import images

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX_AMD = os.path.join(ROOT, "oracle", "_ref", "astc_api_example_amd")
EX_REF = os.path.join(ROOT, "oracle", "_ref", "astc_api_example_ref")


def _need(path):
    if not os.path.exists(path):
        pytest.skip("%s not built (oracle/Makefile `example` needs /root/reference)" % os.path.relpath(path, ROOT))


def test_example_binds_only_exported_symbols(A, built):
    _need(EX_AMD)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", EX_AMD], capture_output=True, text=True, check=True).stdout
    wanted = sorted({line.split()[-1].split("@")[0] for line in undefined.splitlines() if "astcenc_" in line})
    assert wanted, "the example should call into the library"
    exported = subprocess.run(["nm", "-D", "--defined-only", A.LIB_PRODUCT], capture_output=True, text=True, check=True).stdout
    have = {line.split()[-1] for line in exported.splitlines()}
    assert set(wanted) <= have, sorted(set(wanted) - have)
    assert set(wanted) <= set(A.EXPORTS)
    needed = subprocess.run(["readelf", "-d", EX_AMD], capture_output=True, text=True, check=True).stdout
    assert "libastcenc_amd.so" in needed and "libastcenc-none" not in needed


@pytest.mark.gpu
def test_example_program_runs_against_the_gpu_library(tmp_path, A, product):
    from PIL import Image
    _need(EX_AMD)
    img = images.noisy(200, 150, 4)
    img[40:90, 60:140] = (30, 160, 220, 255)           # a flat patch: void-extent blocks too
    src = str(tmp_path / "in.png")
    Image.fromarray(img, "RGBA").save(src)
    outs = {}
    for tag, exe in (("amd", EX_AMD), ("ref", EX_REF)):
        if not os.path.exists(exe):
            continue
        dst = str(tmp_path / ("out_%s.png" % tag))
        run = subprocess.run([exe, src, dst], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, (tag, run.stdout, run.stderr)
        outs[tag] = np.array(Image.open(dst))
    assert outs["amd"].shape == (150, 200, 4)
    # what the example did, redone through the binding: compress + decompress with the same settings
    blocks = product.compress(img, (6, 6), A.PRE_MEDIUM)
    assert np.array_equal(outs["amd"], product.decompress(blocks, 200, 150, (6, 6)))
    if "ref" in outs:
        assert np.array_equal(outs["amd"], outs["ref"])
