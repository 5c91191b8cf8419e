# SPDX-License-Identifier: Apache-2.0
"""TEST INFRASTRUCTURE ONLY -- where the checker libraries live.

  oracle/_ref/libastcenc-none.so    the reference encoder (scalar build) compiled by oracle/Makefile: the authority
  oracle/_ref/libastcenc-avx2.so    the reference's AVX2 build: the timed CPU baseline (byte-identical by its invariance mode)
  oracle/_ref/libastcenc-avx2-gathers.so   the same with ASTCENC_X86_GATHERS=1 (the reference's x86 default); bench.py times both
  oracle/_ref/libastcenc-avx2-lto.so       the AVX2 build with -flto; bench.py reports the fastest of the three
  oracle/_ref/libastcenc-diag.so    the reference with ASTCENC_DIAGNOSTICS (-dtrace JSON): stage-level error oracle
  oracle/emu/_build/libastcenc_emu.so   sequential CPU build of the kernel source (a debugging aid, not independent)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product binding
(astc-encoder_amd/python/astcenc_amd.py) knows nothing about these paths.
"""
import os

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB_REF_NONE = os.path.join(REPO, "oracle", "_ref", "libastcenc-none.so")
LIB_REF_AVX2 = os.path.join(REPO, "oracle", "_ref", "libastcenc-avx2.so")
LIB_REF_AVX2_GATHERS = os.path.join(REPO, "oracle", "_ref", "libastcenc-avx2-gathers.so")
LIB_REF_AVX2_LTO = os.path.join(REPO, "oracle", "_ref", "libastcenc-avx2-lto.so")   # link-time optimised (the reference's release setting for its CLI)
LIB_REF_DIAG = os.path.join(REPO, "oracle", "_ref", "libastcenc-diag.so")
LIB_EMU = os.path.join(REPO, "oracle", "emu", "_build", "libastcenc_emu.so")
# ... with the lanes of every lane loop in reverse order (oracle/emu/Makefile: `make reverse`), the lane-order race check
LIB_EMU_REVERSE = os.path.join(REPO, "oracle", "emu", "_build", "libastcenc_emu_reverse.so")
