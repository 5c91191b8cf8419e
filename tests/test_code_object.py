"""Static properties of the kernels inside the shipped library (no GPU needed): the code objects are pulled out of
libastcenc_amd.so's fat binary and their kernel descriptors read with the ROCm LLVM tools.

The compression kernels are required to run without scratch memory: every scratch store of a spilled register is
256 B of HBM write traffic per wavefront (round 1: 177 such stores per block, 136x the algorithmic traffic)."""
import os, re, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "astc-encoder_amd", "python"))
import astcenc_amd as A  # noqa: E402

BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def kernel_descriptors(lib, tmp):
    """{kernel name: {descriptor field: int}} of every gfx950 kernel in the library."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)] + [len(data)]
    kernels = {}
    for n in range(len(starts) - 1):
        bundle, co = os.path.join(tmp, "b%d.bin" % n), os.path.join(tmp, "k%d.co" % n)
        open(bundle, "wb").write(data[starts[n]:starts[n + 1]])
        subprocess.run([BUNDLER, "--unbundle", "--type=o", "--input=" + bundle, "--targets=" + TARGET, "--output=" + co], check=True)
        notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True, check=True).stdout
        # one YAML map per kernel under amdhsa.kernels; the fields of a kernel precede the next "- .agpr_count" / "- .args"
        for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            kernels[name] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", block, re.M)}
    return kernels


@pytest.mark.skipif(not (os.path.exists(A.LIB_PRODUCT) and os.path.exists(BUNDLER) and os.path.exists(READELF) and shutil.which("objcopy")),
                    reason="needs the built product library and the ROCm LLVM tools")
def test_kernel_descriptors(tmp_path):
    k = kernel_descriptors(A.LIB_PRODUCT, str(tmp_path))
    by_short = {re.sub(r"^_ZN5astcd\d+", "", n): d for n, d in k.items()}
    # four builds of the compression kernel: LDR / HDR coders x footprints of at most 64 texels ("64") / larger ones
    ldr = next(d for n, d in by_short.items() if n.startswith("astc_compress_blocks_ldr64"))
    hdr = next(d for n, d in by_short.items() if n.startswith("astc_compress_blocks_hdr64"))
    for name in ("astc_compress_blocks_ldrEP", "astc_compress_blocks_hdrEP"):
        big = next(d for n, d in by_short.items() if n.startswith(name))
        assert big["private_segment_fixed_size"] == 0 and big["vgpr_spill_count"] == 0 and big["vgpr_count"] <= 128, (name, big)
    # the stages of the compression kernels address LDS from 0 (ctx_make, wave_ctx.h): the kernels must use dynamic LDS only
    for n, d in by_short.items():
        if n.startswith("astc_compress_blocks"):
            assert d["group_segment_fixed_size"] == 0, (n, d)
    # the decoder: no scratch, the LDS record of a run of 32 blocks (DecodeBatch) lets 23 single-wave workgroups share a CU's
    # 160 KB, and the registers must not be what limits the waves (512 / 80 = 6 per SIMD >= 23 / 4)
    dec = next(d for n, d in by_short.items() if n.startswith("astc_decompress_blocks"))
    assert dec["private_segment_fixed_size"] == 0 and dec["vgpr_spill_count"] == 0, dec
    assert dec["group_segment_fixed_size"] <= 7040 and dec["vgpr_count"] <= 80 and dec["max_flat_workgroup_size"] == 64, dec
    assert any(n.startswith("astc_alpha_averages") for n in by_short)
    assert sum(n.startswith("astc_compare_") for n in by_short) == 3
    # the headline kernel: no scratch memory at all, the register budget of four wavefronts per SIMD
    assert ldr["private_segment_fixed_size"] == 0 and ldr["vgpr_spill_count"] == 0, ldr
    assert ldr["vgpr_count"] <= 128 and ldr["max_flat_workgroup_size"] == 64, ldr
    # ... and so does the HDR variant
    assert hdr["private_segment_fixed_size"] == 0 and hdr["vgpr_spill_count"] == 0 and hdr["vgpr_count"] <= 128, hdr
    # ... and the fixed-context builds (what BASELINE configs[1..3] actually run: kernel_ldr_6x6m.hip, kernel_ldr_8x8t.hip,
    # kernel_hdr_6x6m.hip)
    for name in ("astc_compress_blocks_ldr_6x6m", "astc_compress_blocks_ldr_8x8t", "astc_compress_blocks_hdr_6x6m"):
        d = next(d for n, d in by_short.items() if n.startswith(name))
        assert d["private_segment_fixed_size"] == 0 and d["vgpr_spill_count"] == 0, (name, d)
        assert d["vgpr_count"] <= 128 and d["max_flat_workgroup_size"] == 64, (name, d)
