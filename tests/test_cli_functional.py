# SPDX-License-Identifier: Apache-2.0
"""The reference's own command line tool with the drop-in as its codec.

oracle/Makefile (`cli`) compiles the reference's front end -- Source/astcenccli_*.cpp: argument parsing, image file IO,
the .astc container, the error metrics -- from where it lies and links it once to libastcenc_amd.so
(oracle/_ref/astcenc-cli-amd: the link-time swap of INTEGRATION.md section 2) and once to the reference library
(oracle/_ref/astcenc-cli-ref).  Same front end, two back ends: whatever the two write for the same command line must be
the same bytes, and whatever they print on a failing command line the same text.

The cases restate /root/reference/Test/astc_test_functional.py (the reference's functional suite, which needs ImageMagick
for its pixel checks and cannot run here): the known-answer files (:457-503), decompression and round trips (:505-605),
every 2D and 3D block size (:607-655), the presets (:657-673), input and output containers (:675-814), -normal /
-perceptual (:816-857), the swizzles (:859-930), -flip (:932-1014), -cw (:1016-1045), the tuning switches (:1047-1355),
-j (:1357), -silent (:1381), the array input of -zdim (:1672), and its negative tests (:1537-2260), of which the ones
that reach the codec (bad block sizes, bad presets) must produce the reference's error text out of THIS library.  Where
the reference's test checks a pixel colour or an RMSE ordering, the check here is stronger: the output file equals
what the reference library makes of the same command.  Test data: tests/golden/cli_data (copied from the reference's
Test/Data by tests/golden/make_cli_data.py)."""
import filecmp
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI_AMD = os.path.join(ROOT, "oracle", "_ref", "astcenc-cli-amd")
CLI_REF = os.path.join(ROOT, "oracle", "_ref", "astcenc-cli-ref")
DATA = os.path.join(ROOT, "tests", "golden", "cli_data")
TILE = os.path.join(DATA, "Tiles", "ldr.png")
TILE_HDR = os.path.join(DATA, "Tiles", "hdr.exr")
COMPLEX = os.path.join(DATA, "Tiles", "ldr-complex.png")
COMPLEX_HDR = os.path.join(DATA, "Tiles", "hdr-complex.exr")

pytestmark = pytest.mark.gpu


def _need():
    for exe in (CLI_AMD, CLI_REF):
        if not os.path.exists(exe):
            pytest.skip("%s not built (oracle/Makefile `cli` needs /root/reference)" % os.path.relpath(exe, ROOT))


def run(exe, args, cwd):
    return subprocess.run([exe] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=600)


def both(tmp_path, args, outputs=(), expect_ok=True):
    """Run the same command line through both tools, each in its own directory; `outputs`: files (relative names) both must
    have written with identical bytes.  Returns the two completed processes."""
    _need()
    res = {}
    for tag, exe in (("amd", CLI_AMD), ("ref", CLI_REF)):
        d = tmp_path / tag
        d.mkdir(exist_ok=True)
        res[tag] = run(exe, args, str(d))
    a, r = res["amd"], res["ref"]
    assert a.returncode == r.returncode, (args, a.returncode, r.returncode, a.stdout[-500:], a.stderr[-500:])
    if expect_ok:
        assert a.returncode == 0, (args, a.stdout[-800:], a.stderr[-800:])
    for name in outputs:
        fa, fr = tmp_path / "amd" / name, tmp_path / "ref" / name
        assert fa.exists() and fr.exists(), (args, name)
        assert filecmp.cmp(str(fa), str(fr), shallow=False), "%s differs between the two back ends for %s" % (name, args)
    return a, r


# ---- known answers (astc_test_functional.py:457-503) --------------------------------------------------------------
@pytest.mark.parametrize("switch,src,kat", [("-cl", "LDR-A-1x1.png", "LDR-A-1x1.astc"), ("-cs", "LDRS-A-1x1.png", "LDRS-A-1x1.astc"),
                                             ("-ch", "HDR-A-1x1.exr", "HDR-A-1x1.astc"), ("-cH", "HDR-A-1x1.exr", "HDR-A-1x1.astc")])
def test_known_answer_files(tmp_path, switch, src, kat):
    both(tmp_path, [switch, os.path.join(DATA, src), "out.astc", "6x6", "-exhaustive"], ["out.astc"])
    assert filecmp.cmp(str(tmp_path / "amd" / "out.astc"), os.path.join(DATA, kat), shallow=False)


# ---- decompression and round trips (:505-605) -----------------------------------------------------------------------
@pytest.mark.parametrize("switch,kat,out", [("-dl", "LDR-A-1x1.astc", "out.png"), ("-ds", "LDRS-A-1x1.astc", "out.png"),
                                             ("-dh", "HDR-A-1x1.astc", "out.exr"), ("-dH", "HDR-A-1x1.astc", "out.exr"),
                                             ("-dl", "Tiles/ldr.astc", "out.tga"), ("-dh", "Tiles/hdr.astc", "out.hdr")])
def test_decompress(tmp_path, switch, kat, out):
    both(tmp_path, [switch, os.path.join(DATA, kat), out], [out])


@pytest.mark.parametrize("switch,src,out", [("-tl", "LDR-A-1x1.png", "out.png"), ("-ts", "LDRS-A-1x1.png", "out.png"),
                                             ("-th", "HDR-A-1x1.exr", "out.exr"), ("-tH", "HDR-A-1x1.exr", "out.exr")])
def test_roundtrip(tmp_path, switch, src, out):
    both(tmp_path, [switch, os.path.join(DATA, src), out, "6x6", "-exhaustive"], [out])


# ---- block sizes and presets (:607-673) -----------------------------------------------------------------------------
@pytest.mark.parametrize("block", ["4x4", "5x4", "5x5", "6x5", "6x6", "8x5", "8x6", "10x5", "10x6", "8x8", "10x8", "10x10", "12x10", "12x12",
                                   "3x3x3", "4x3x3", "4x4x3", "4x4x4", "5x4x4", "5x5x4", "5x5x5", "6x5x5", "6x6x5", "6x6x6"])
def test_valid_block_sizes(tmp_path, block):
    both(tmp_path, ["-cl", COMPLEX, "out.astc", block, "-medium"], ["out.astc"])


@pytest.mark.parametrize("preset", ["-fastest", "-fast", "-medium", "-thorough", "-verythorough", "-exhaustive"])
def test_valid_presets(tmp_path, preset):
    both(tmp_path, ["-tl", COMPLEX, "out.png", "4x4", preset], ["out.png"])
    both(tmp_path, ["-ch", COMPLEX_HDR, "out.astc", "6x6", preset], ["out.astc"])


# ---- containers (:675-814) ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ext", ["bmp", "dds", "jpg", "ktx", "png", "tga"])
def test_ldr_input_formats(tmp_path, ext):
    both(tmp_path, ["-tl", os.path.join(DATA, "Tiles", "ldr." + ext), "out.png", "4x4", "-fast"], ["out.png"])


@pytest.mark.parametrize("ext", ["exr", "hdr"])
def test_hdr_input_formats(tmp_path, ext):
    both(tmp_path, ["-th", os.path.join(DATA, "Tiles", "hdr." + ext), "out.exr", "4x4", "-fast"], ["out.exr"])


@pytest.mark.parametrize("ext", ["bmp", "dds", "ktx", "png", "tga"])
def test_ldr_output_formats(tmp_path, ext):
    both(tmp_path, ["-tl", TILE, "out." + ext, "4x4", "-fast"], ["out." + ext])


@pytest.mark.parametrize("ext", ["dds", "exr", "hdr", "ktx"])
def test_hdr_output_formats(tmp_path, ext):
    both(tmp_path, ["-th", TILE_HDR, "out." + ext, "4x4", "-fast"], ["out." + ext])


@pytest.mark.parametrize("switch,src,ext", [("-cl", TILE, "astc"), ("-cl", TILE, "ktx"), ("-ch", TILE_HDR, "astc"), ("-ch", TILE_HDR, "ktx")])
def test_compressed_output_formats(tmp_path, switch, src, ext):
    both(tmp_path, [switch, src, "out." + ext, "4x4", "-fast"], ["out." + ext])


# ---- options that change what the codec is asked to do (:816-1355) ----------------------------------------------------
OPTION_CASES = [
    ["-normal"], ["-normal", "-perceptual"], ["-rgbm", "8"], ["-perceptual"], ["-decode_unorm8"],
    ["-esw", "rgba"], ["-esw", "g0r1"], ["-esw", "rrrg"], ["-dsw", "g0r1"], ["-dsw", "rrrg"], ["-esw", "gbar", "-dsw", "argb"],
    ["-esw", "rrrg", "-ssw", "ra"], ["-yflip"],
    ["-cw", "10", "1", "1", "1"], ["-cw", "1", "10", "1", "1"], ["-cw", "1", "1", "10", "1"], ["-cw", "1", "1", "1", "10"],
    ["-a", "1"], ["-a", "2"],
    ["-partitioncountlimit", "1"], ["-2partitionindexlimit", "1"], ["-3partitionindexlimit", "1"], ["-4partitionindexlimit", "1"],
    ["-blockmodelimit", "25"], ["-refinementlimit", "1"], ["-candidatelimit", "1"], ["-dblimit", "10"],
    ["-2partitionlimitfactor", "1.0"], ["-3partitionlimitfactor", "1.0"], ["-2planelimitcorrelation", "0.1"],
    ["-2partitioncandidatelimit", "1"], ["-3partitioncandidatelimit", "1"], ["-4partitioncandidatelimit", "1"],
    ["-j", "1"], ["-j", "3"], ["-silent"], ["-repeats", "2"],
]


@pytest.mark.parametrize("options", OPTION_CASES, ids=[" ".join(o) for o in OPTION_CASES])
def test_options(tmp_path, options):
    both(tmp_path, ["-tl", COMPLEX, "out.png", "4x4", "-medium"] + options, ["out.png"])
    both(tmp_path, ["-cl", COMPLEX, "out.astc", "6x6", "-thorough"] + options, ["out.astc"])


def test_flip_on_compression_and_decompression(tmp_path):
    # (:932-1014) compress flipped, decompress flipped, round trip with both
    both(tmp_path, ["-cl", TILE, "out.astc", "4x4", "-fast", "-yflip"], ["out.astc"])
    shutil.copy(str(tmp_path / "amd" / "out.astc"), str(tmp_path / "flipped.astc"))
    both(tmp_path, ["-dl", str(tmp_path / "flipped.astc"), "out.png", "-yflip"], ["out.png"])
    both(tmp_path, ["-tl", TILE, "rt.png", "4x4", "-fast", "-yflip"], ["rt.png"])


def test_array_input_as_volume(tmp_path):
    # (:1672-1698: the slices ldr_0.png, ldr_1.png are found from the stem and -zdim)
    both(tmp_path, ["-cl", os.path.join(DATA, "Tiles", "ldr.png"), "out.astc", "4x4x4", "-fast", "-zdim", "2"], ["out.astc"])
    both(tmp_path, ["-cl", os.path.join(DATA, "Tiles", "ldr.png"), "out.astc", "6x6", "-fast", "-zdim", "2"], expect_ok=False)


def test_quality_report_is_the_same_text(tmp_path):
    """The `-tl` report (PSNR lines of the reference's compute_error_metrics on what each back end decoded)."""
    a, r = both(tmp_path, ["-tl", COMPLEX, "out.png", "6x6", "-medium"], ["out.png"])
    pick = lambda text: [l for l in text.splitlines() if "PSNR" in l or "LogRMSE" in l]
    assert pick(a.stdout) and pick(a.stdout) == pick(r.stdout)
    a, r = both(tmp_path, ["-th", COMPLEX_HDR, "out.exr", "6x6", "-medium"], ["out.exr"])
    assert pick(a.stdout) and pick(a.stdout) == pick(r.stdout)


# ---- negative tests (:1537-2260): same exit code, same message --------------------------------------------------------
NEGATIVE = [
    ["-cl", TILE, "out.astc", "4x7", "-fast"], ["-cl", TILE, "out.astc", "3x3", "-fast"], ["-cl", TILE, "out.astc", "4x4x7", "-fast"],
    ["-cl", TILE, "out.astc", "7x7x7", "-fast"], ["-cl", TILE, "out.astc", "4x4x", "-fast"], ["-cl", TILE, "out.astc", "0x0", "-fast"],
    ["-cl", TILE, "out.astc", "4x4", "-superfast"], ["-cl", TILE, "out.astc", "4x4", "101"], ["-cl", TILE, "out.astc", "4x4", "-1"],
    ["-cl", TILE, "out.astc", "4x4", "-fast", "-unknown"], ["-cl", TILE, "out.astc", "4x4"], ["-cl", TILE], ["-cl"],
    ["-cl", "missing.png", "out.astc", "4x4", "-fast"], ["-cl", os.path.join(DATA, "empty.unk"), "out.astc", "4x4", "-fast"],
    ["-cl", TILE, "./nodir/out.astc", "4x4", "-fast"], ["-cl", TILE, "out.xyz", "4x4", "-fast"],
    ["-tl", TILE, "out.png", "4x7", "-fast"], ["-tl", TILE, "out.png", "4x4", "-superfast"], ["-tl", TILE, "out.png", "4x4", "-fast", "-unknown"],
    ["-tl", TILE, "out.png"], ["-tl", "missing.png", "out.png", "4x4", "-fast"], ["-dl", os.path.join(DATA, "Tiles", "ldr.astc")], ["-dl"],
    ["-dl", os.path.join(DATA, "Tiles", "ldr.astc"), "./nodir/out.png"],
    ["-cl", TILE, "out.astc", "4x4", "-fast", "-a"], ["-cl", TILE, "out.astc", "4x4", "-fast", "-cw", "1", "1", "1"],
    ["-cl", TILE, "out.astc", "4x4", "-fast", "-partitioncountlimit"], ["-cl", TILE, "out.astc", "4x4", "-fast", "-blockmodelimit"],
    ["-cl", TILE, "out.astc", "4x4", "-fast", "-refinementlimit"], ["-cl", TILE, "out.astc", "4x4", "-fast", "-dblimit"],
    ["-cl", TILE, "out.astc", "4x4", "-fast", "-esw"], ["-cl", TILE, "out.astc", "4x4", "-fast", "-esw", "rgb"],
    ["-cl", TILE, "out.astc", "4x4", "-fast", "-esw", "rgbq"], ["-cl", TILE, "out.astc", "4x4", "-fast", "-ssw"],
    ["-cl", TILE, "out.astc", "4x4", "-fast", "-ssw", "rgbaa"], ["-cl", TILE, "out.astc", "4x4", "-fast", "-ssw", "q"],
    ["-dl", os.path.join(DATA, "Tiles", "ldr.astc"), "out.png", "-dsw"], ["-dl", os.path.join(DATA, "Tiles", "ldr.astc"), "out.png", "-dsw", "rgbq"],
    ["-ch", TILE_HDR, "out.astc", "4x4", "-fast", "-mpsnr", "10"],
    ["-dl", os.path.join(DATA, "negative_magic.astc"), "out.png"], ["-dl", os.path.join(DATA, "negative_huge.astc"), "out.png"],
    ["-dl", os.path.join(DATA, "negative_overflow.astc"), "out.png"], ["-dl", os.path.join(DATA, "negative_short.astc"), "out.png"],
    ["-dl", os.path.join(DATA, "negative_block_size.astc"), "out.png"],
]


@pytest.mark.parametrize("args", NEGATIVE, ids=[" ".join(os.path.basename(str(a)) for a in n) for n in NEGATIVE])
def test_negative(tmp_path, args):
    a, r = both(tmp_path, args, expect_ok=False)
    assert a.returncode != 0, args
    # the tools print their own name nowhere in these messages: the text must be the same
    assert a.stdout == r.stdout and a.stderr == r.stderr, (args, a.stdout[-400:], r.stdout[-400:], a.stderr[-400:], r.stderr[-400:])
