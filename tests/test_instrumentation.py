# SPDX-License-Identifier: Apache-2.0
"""The instrumentation build (astc-encoder_amd/libastcenc_amd_trace.so: -DASTC_TRACE -DASTC_DUPSTAGE, never the product):
tools/gpu_stage_counts.sh measures a stage's dynamic instruction count as the counter difference between a run with that
stage executed twice (DUP_STAGE, wave_ctx.h; ASTC_DUP_STAGE in the environment picks it) and a plain run.  That is only
meaningful if the doubled run does the same work otherwise -- i.e. if it produces the same bytes.  Every stage id, on a
fixed-context build (6x6 -medium) and on a generic one (5x5 -thorough)."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_TRACE = os.path.join(ROOT, "astc-encoder_amd", "libastcenc_amd_trace.so")

SCRIPT = r"""
import sys, hashlib, os
sys.path.insert(0, %r)
import numpy as np, torch
torch.zeros(1, device="cuda:0")
import astcenc_amd as A
lib = A.Library(sys.argv[1])
img = A.synthetic_image(126, 90, 11)
for stage in range(int(sys.argv[2])):
    # (read when a context is created, astcenc_entry.cpp; lib.compress() creates one per call)
    os.environ["ASTC_DUP_STAGE"] = str(stage)
    print(stage, " ".join(hashlib.sha256(np.asarray(lib.compress(img, (b, b), q)).tobytes()).hexdigest() for b, q in ((6, A.PRE_MEDIUM), (5, A.PRE_THOROUGH))))
""" % os.path.join(ROOT, "astc-encoder_amd", "python")


def _digests(lib_path, stages):
    r = subprocess.run([sys.executable, "-c", SCRIPT, lib_path, str(stages)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = [l.split() for l in r.stdout.strip().split("\n") if l and l[0].isdigit()]
    assert len(rows) == stages
    return {int(row[0]): row[1:] for row in rows}


@pytest.mark.gpu
def test_every_doubled_stage_leaves_the_bytes_unchanged(built, A):
    assert os.path.exists(LIB_TRACE), "instrumentation build missing: __graft_entry__.build() makes it"
    want = _digests(A.LIB_PRODUCT, 1)[0]
    got = _digests(LIB_TRACE, 32)
    for stage, digests in got.items():
        assert digests == want, "stage %d doubled changes the output" % stage
