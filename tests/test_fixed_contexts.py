# SPDX-License-Identifier: Apache-2.0
"""Fixed-context builds of the compression kernel (kernel_ldr_6x6m.hip, kernel_ldr_8x8t.hip, kernel_hdr_6x6m.hip): each is
compiled for ONE context -- its LdsLayout, DeviceConfig and TableRoot are compile-time constants taken from
csrc/fixed_contexts.inc -- and the backend uses it only when the live context equals those records byte for byte.

CPU: the committed fixed_contexts.inc is what the host code computes today (regenerated through the sequential build).
GPU: the three BASELINE contexts do get their fixed build, every other context (another preset, a flag, a channel weight,
     a partition limit) the generic one, ASTCENC_AMD_KERNEL=generic switches the fixed builds off, and both builds give the
     reference's bytes on the same image."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixed_contexts_inc_is_current(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fixed_contexts.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]


def _kernel_name(lib, A, profile, block, quality, flags=0, tweak=None):
    err, cfg = lib.config_init(profile, block, block, 1, quality, flags)
    assert err == 0
    if tweak:
        tweak(cfg)
    err, ctx = lib.context_alloc(cfg, 1)
    assert err == 0
    try:
        return lib.lib.astcenc_amd_context_kernel_name(ctx).decode()
    finally:
        lib.context_free(ctx)


@pytest.mark.gpu
def test_baseline_contexts_get_their_fixed_build(product, A):
    assert _kernel_name(product, A, A.PRF_LDR, 6, A.PRE_MEDIUM) == "astc_compress_blocks_ldr_6x6m"
    assert _kernel_name(product, A, A.PRF_LDR, 8, A.PRE_THOROUGH) == "astc_compress_blocks_ldr_8x8t"
    assert _kernel_name(product, A, A.PRF_HDR, 6, A.PRE_MEDIUM) == "astc_compress_blocks_hdr_6x6m"


@pytest.mark.gpu
def test_other_contexts_get_the_generic_build(product, A):
    assert _kernel_name(product, A, A.PRF_LDR, 6, A.PRE_FAST) == "astc_compress_blocks_ldr64"
    assert _kernel_name(product, A, A.PRF_LDR, 6, A.PRE_THOROUGH) == "astc_compress_blocks_ldr64"
    assert _kernel_name(product, A, A.PRF_LDR_SRGB, 6, A.PRE_MEDIUM) == "astc_compress_blocks_ldr64"
    assert _kernel_name(product, A, A.PRF_LDR, 8, A.PRE_MEDIUM) == "astc_compress_blocks_ldr64"
    assert _kernel_name(product, A, A.PRF_LDR, 5, A.PRE_MEDIUM) == "astc_compress_blocks_ldr64"
    assert _kernel_name(product, A, A.PRF_LDR, 10, A.PRE_MEDIUM) == "astc_compress_blocks_ldr"
    assert _kernel_name(product, A, A.PRF_HDR, 8, A.PRE_MEDIUM) == "astc_compress_blocks_hdr64"
    assert _kernel_name(product, A, A.PRF_HDR_RGB_LDR_A, 6, A.PRE_MEDIUM) == "astc_compress_blocks_hdr64"
    # a flag, a channel weight, a tuning field: anything that changes one of the three records
    assert _kernel_name(product, A, A.PRF_LDR, 6, A.PRE_MEDIUM, flags=A.FLG_USE_ALPHA_WEIGHT) == "astc_compress_blocks_ldr64"

    def heavier_red(cfg):
        cfg.cw_r_weight = 2.0
    assert _kernel_name(product, A, A.PRF_LDR, 6, A.PRE_MEDIUM, tweak=heavier_red) == "astc_compress_blocks_ldr64"

    def two_partitions(cfg):
        cfg.tune_partition_count_limit = 2
    assert _kernel_name(product, A, A.PRF_LDR, 6, A.PRE_MEDIUM, tweak=two_partitions) == "astc_compress_blocks_ldr64"


SCRIPT = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, torch
torch.zeros(1, device="cuda:0")
import astcenc_amd as A
lib = A.Library(A.LIB_PRODUCT)
out = []
for prof, b, q, hdr in ((A.PRF_LDR, 6, A.PRE_MEDIUM, 0), (A.PRF_LDR, 8, A.PRE_THOROUGH, 0), (A.PRF_HDR, 6, A.PRE_MEDIUM, 1)):
    err, cfg = lib.config_init(prof, b, b, 1, q, 0); assert err == 0
    err, ctx = lib.context_alloc(cfg, 1); assert err == 0
    name = lib.lib.astcenc_amd_context_kernel_name(ctx).decode()
    lib.context_free(ctx)
    img = A.synthetic_hdr_image(250, 190, 3) if hdr else A.synthetic_image(250, 190, 3)
    data = lib.compress(img, (b, b), q, profile=prof)
    out.append("%%s %%s" %% (name, hashlib.sha256(np.asarray(data).tobytes()).hexdigest()))
print("\n".join(out))
""" % os.path.join(ROOT, "astc-encoder_amd", "python")


def _run_script(extra_env):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return [line.split() for line in r.stdout.strip().split("\n")[-3:]]


@pytest.mark.gpu
def test_fixed_and_generic_builds_give_the_same_bytes(product, ref, A):
    """The same three images through the fixed builds and, in a second process with ASTCENC_AMD_KERNEL=generic, through the
    generic ones: different kernels, the same bytes -- and those are the reference's."""
    fixed = _run_script({})
    generic = _run_script({"ASTCENC_AMD_KERNEL": "generic"})
    assert [n for n, _ in fixed] == ["astc_compress_blocks_ldr_6x6m", "astc_compress_blocks_ldr_8x8t", "astc_compress_blocks_hdr_6x6m"]
    assert [n for n, _ in generic] == ["astc_compress_blocks_ldr64", "astc_compress_blocks_ldr64", "astc_compress_blocks_hdr64"]
    assert [h for _, h in fixed] == [h for _, h in generic]
    import hashlib
    for (prof, b, q, hdr), (_, digest) in zip(((A.PRF_LDR, 6, A.PRE_MEDIUM, 0), (A.PRF_LDR, 8, A.PRE_THOROUGH, 0), (A.PRF_HDR, 6, A.PRE_MEDIUM, 1)), fixed):
        img = A.synthetic_hdr_image(250, 190, 3) if hdr else A.synthetic_image(250, 190, 3)
        want = ref.compress(img, (b, b), q, profile=prof)
        assert hashlib.sha256(np.asarray(want).tobytes()).hexdigest() == digest
