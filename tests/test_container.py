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


def test_ktx_header_layout_and_round_trip(tmp_path, A):
    """Field-by-field against the reference writer (store_ktx_compressed_image,
    astcenccli_image_load_store.cpp:1396-1437): 12-byte magic, 13 little-endian u32s, u32 length, data."""
    import struct
    rng = np.random.default_rng(2)
    blocks = rng.integers(0, 256, size=16 * 4 * 3, dtype=np.uint8)
    p = tmp_path / "a.ktx"
    A.write_ktx(str(p), blocks, 22, 14, (6, 5), srgb=True)
    raw = p.read_bytes()
    assert raw[:12] == bytes([0xAB, 0x4B, 0x54, 0x58, 0x20, 0x31, 0x31, 0xBB, 0x0D, 0x0A, 0x1A, 0x0A])
    assert struct.unpack_from("<13I", raw, 12) == (0x04030201, 0, 1, 0, 0x93D3, 0x1908, 22, 14, 0, 0, 1, 1, 0)
    assert struct.unpack_from("<I", raw, 64)[0] == blocks.size and raw[68:] == blocks.tobytes()
    got, w, h, d, block, srgb = A.read_ktx(str(p))
    assert (w, h, d, block, srgb) == (22, 14, 1, (6, 5, 1), True) and (got == blocks).all()

    # a volume with a 3D footprint uses the _OES enums and a non-zero pixelDepth
    vol_blocks = rng.integers(0, 256, size=16 * 2 * 2 * 3, dtype=np.uint8)
    A.write_ktx(str(p), vol_blocks, 8, 7, (4, 4, 3), depth=9)
    raw = p.read_bytes()
    assert struct.unpack_from("<13I", raw, 12)[4:9] == (0x93C2, 0x1908, 8, 7, 9)
    got, w, h, d, block, srgb = A.read_ktx(str(p))
    assert (w, h, d, block, srgb) == (8, 7, 9, (4, 4, 3), False) and (got == vol_blocks).all()

    # opposite byte order is accepted, as the reference does (ktx_header_switch_endianness)
    swapped = raw[:12] + b"".join(raw[i:i + 4][::-1] for i in range(12, 68, 4)) + raw[68:]
    p.write_bytes(swapped)
    got, w, h, d, block, srgb = A.read_ktx(str(p))
    assert (w, h, d, block) == (8, 7, 9, (4, 4, 3)) and (got == vol_blocks).all()

    for bad in (raw[:40], b"\x00" + raw[1:], raw[:28] + struct.pack("<I", 0x8058) + raw[32:], raw[:-1],
                raw[:16] + struct.pack("<I", 0x1401) + raw[20:]):
        p.write_bytes(bad)
        with pytest.raises(ValueError):
            A.read_ktx(str(p))
    assert A.ktx_gl_format((12, 12)) == 0x93BD and A.ktx_gl_format((6, 6, 6), True) == 0x93E9
