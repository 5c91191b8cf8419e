# SPDX-License-Identifier: Apache-2.0
"""wave.h turns the reference's compare-select min / max / clamp into single hardware instructions (v_min_f32, v_max_f32,
v_med3_f32 through inline asm) where the two agree bit for bit; tools/minmax_semantics.hip states exactly what is relied
on -- the -0 / NaN cases under the floating-point mode HIP kernels run in -- and checks it on the hardware.  A toolchain
or mode change that breaks the equivalence fails here instead of as a byte mismatch somewhere in the sweeps."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_min_max_med3_semantics_on_the_hardware(tmp_path):
    exe = str(tmp_path / "minmax_semantics")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(ROOT, "tools", "minmax_semantics.hip")],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "violations of the contract wave.h relies on: 0" in r.stdout
