# SPDX-License-Identifier: Apache-2.0
"""The shipped library: exports exactly the C ABI that include/*.h declares, loads on a box with no
GPU, and refuses to create a compression context there instead of falling back to any CPU path."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(astcenc_[a-z0-9_]+)\s*\(", text))


def test_exports_every_declared_symbol_and_nothing_else(product, A):
    declared = _declared("astcenc.h") | _declared("astcenc_amd.h")
    assert declared == set(A.EXPORTS) | set(A.EXPORTS_AMD)
    handle = ctypes.CDLL(A.LIB_PRODUCT)
    for sym in declared:
        assert getattr(handle, sym) is not None
    # EVERY defined dynamic symbol, whatever its type (T, D, B, V, W ...): the library is linked with a version script
    # (astc-encoder_amd/exports.map), so kernel handle objects, __hip_cuid_* markers and weak libstdc++ instantiations
    # stay local -- the reference's library exports its functions and nothing else
    out = subprocess.run(["nm", "-D", "--defined-only", A.LIB_PRODUCT], capture_output=True, text=True, check=True).stdout
    exported = {}
    for line in out.splitlines():
        parts = line.split()
        if len(parts) >= 3:
            exported[parts[-1]] = parts[-2]
    assert set(exported) == declared, (sorted(set(exported) - declared), sorted(declared - set(exported)))
    assert all(kind == "T" for kind in exported.values()), exported


def test_product_does_not_link_the_oracle(A):
    out = subprocess.run(["ldd", A.LIB_PRODUCT], capture_output=True, text=True).stdout
    assert "oracle" not in out and "astcenc-none" not in out and "emu" not in out, out
    strings = subprocess.run(["strings", "-n", "6", A.LIB_PRODUCT], capture_output=True, text=True).stdout if _have("strings") else ""
    # (the library carries its device source as text for the run-time builds, csrc/kernel_jit.cpp: comment lines of that
    #  source may name the debugging build; nothing else may)
    hits = [l for l in strings.splitlines() if ("libastcenc_emu" in l or "oracle/" in l) and not l.lstrip().startswith(("//", "*", "/*"))]
    assert not hits, hits


def _have(tool):
    return subprocess.run(["which", tool], capture_output=True).returncode == 0


def test_no_device_means_no_context(product, A):
    """Without a HIP device context_alloc must fail loudly (no CPU fallback exists in the product)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU; covered by the gpu tests")
    err, cfg = product.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 0)
    assert err == A.SUCCESS                      # config_init is pure host arithmetic
    err, ctx = product.context_alloc(cfg, 1)
    assert err != A.SUCCESS and not ctx.value
    assert product.error_string(err) is not None


def test_diagnostics_go_to_the_callback_not_to_stderr(product, A, capfd):
    """The library prints nothing by itself: the reason behind an error code reaches the
    application through astcenc_amd_set_log_callback only."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU: context_alloc succeeds and has nothing to report")
    lines = []
    CB = ctypes.CFUNCTYPE(None, ctypes.c_char_p)
    cb = CB(lambda msg: lines.append(msg.decode()))
    product.lib.astcenc_amd_set_log_callback.restype = None
    product.lib.astcenc_amd_set_log_callback.argtypes = [CB]
    err, cfg = product.config_init(A.PRF_LDR, 6, 6, 1, A.PRE_MEDIUM, 0)
    err, ctx = product.context_alloc(cfg, 1)              # no callback installed: silent
    assert err != A.SUCCESS
    assert "astcenc_amd" not in capfd.readouterr().err
    product.lib.astcenc_amd_set_log_callback(cb)
    try:
        err, ctx = product.context_alloc(cfg, 1)
        assert err != A.SUCCESS
        assert any("no HIP device" in l for l in lines), lines
        assert "astcenc_amd" not in capfd.readouterr().err
    finally:
        product.lib.astcenc_amd_set_log_callback(CB(0))
