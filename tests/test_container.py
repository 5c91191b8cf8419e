# SPDX-License-Identifier: Apache-2.0
"""The .astc container helpers of the binding module (ref: Docs/FileFormat.md; the header bytes of the
reference's Test/Data/*-1x1.astc files are quoted)."""
import numpy as np
import pytest


def test_astc_header_matches_reference_files(tmp_path, A):
    # Test/Data/LDR-A-1x1.astc: 13 ab a1 5c | 06 06 01 | 01 00 00 | 01 00 00 | 01 00 00 | 16-byte block
    payload = bytes.fromhex("fcfdffffffffffff2b2badad0000ffff")
    p = tmp_path / "a.astc"
    A.write_astc(str(p), np.frombuffer(payload, dtype=np.uint8), 1, 1, (6, 6))
    assert p.read_bytes().hex() == "13aba15c060601010000010000010000" + payload.hex()
    blocks, w, h, d, block = A.read_astc(str(p))
    assert (w, h, d, block) == (1, 1, 1, (6, 6, 1)) and blocks.tobytes() == payload


def test_astc_round_trip_and_negative_cases(tmp_path, A):
    rng = np.random.default_rng(1)
    blocks = rng.integers(0, 256, size=16 * 7 * 5, dtype=np.uint8)
    p = tmp_path / "b.astc"
    A.write_astc(str(p), blocks, 40, 22, (6, 5))
    got, w, h, d, block = A.read_astc(str(p))
    assert (w, h, d, block) == (40, 22, 1, (6, 5, 1)) and (got == blocks).all()
    raw = p.read_bytes()
    for bad in (raw[:15], b"\x00" + raw[1:], raw[:4] + b"\x00" + raw[5:], raw[:-1], raw[:7] + b"\x00\x00\x00" + raw[10:]):
        q = tmp_path / "bad.astc"
        q.write_bytes(bad)
        with pytest.raises(ValueError):
            A.read_astc(str(q))
