# SPDX-License-Identifier: Apache-2.0
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_libs as O  # noqa: E402  (checker libraries: test infrastructure)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Run-time specialised kernel builds (csrc/kernel_jit.cpp): off for the session unless a test asks for them
    # (tests/test_jit.py sets the mode and a cache directory of its own), so that which build a context launches -- asserted
    # by name in tests/test_fixed_contexts.py -- does not depend on what an earlier test left in a disk cache, and the
    # user's ~/.cache is never written by a test run.
    os.environ.setdefault("ASTCENC_AMD_JIT", "off")
    os.environ.setdefault("ASTCENC_AMD_CACHE_DIR", "")


def _have(path):
    return os.path.exists(path)


@pytest.fixture(scope="session")
def A():
    import astcenc_amd
    return astcenc_amd


@pytest.fixture(scope="session")
def built():
    """Build product + emulator (+ oracle when the reference tree is present) once per session."""
    import astcenc_amd as A
    if not (_have(A.LIB_PRODUCT) and _have(O.LIB_EMU) and (_have(O.LIB_REF_NONE) or not os.path.isdir("/root/reference"))):
        import __graft_entry__
        __graft_entry__.build()
    return True


@pytest.fixture(scope="session")
def ref(built, A):
    """The real reference encoder (oracle/_ref). Built here from /root/reference; travels prebuilt to the GPU box."""
    if not _have(O.LIB_REF_NONE):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    return A.Library(O.LIB_REF_NONE)


@pytest.fixture(scope="session")
def emu(built, A):
    return A.Library(O.LIB_EMU)


@pytest.fixture(scope="session")
def product(built, A):
    # torch ships its own HIP runtime under the soname the library links: imported first, both share that
    # one runtime; tests that keep data in HBM through torch rely on this order
    import torch
    if torch.cuda.is_available():
        torch.zeros(1, device="cuda:0")
    return A.Library(A.LIB_PRODUCT)
