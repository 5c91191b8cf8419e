#!/usr/bin/env python3
"""Fetch the reference's image corpus for the parity / throughput tests that run on the GPU box.

/root/reference does not exist there, and 28 MB of third-party images do not belong in this repository's history, so --
like oracle/_ref, which is compiled from the reference's sources where they lie -- the images are COPIED from where
they lie (/root/reference/Test/Images/{Small,Khronos,HDRIHaven}) into tests/corpus/_images/ (git-ignored, not
gpurun-ignored: the copy travels to the GPU box with the working tree).  What IS committed: this script and
tests/corpus/manifest.json -- every image's path, SHA-256, colour profile / format / flags as the reference's test
harness derives them from the file name (/root/reference/Test/testlib/testset.py:117-190), and the reference's own
recorded PSNR per block size and preset (Test/Images/<set>/astc_reference-5.0-avx2_<preset>_results.csv, the values
/root/reference/Test/astc_test_image.py:45-47 gates on).  __graft_entry__.build() runs this when the reference tree
is present."""
import csv, hashlib, json, os, shutil, sys
REF = "/root/reference/Test/Images"
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, "_images")
SETS = ("Small", "Khronos", "HDRIHaven")
EXTS = (".jpg", ".png", ".tga", ".dds", ".hdr", ".ktx")            # testset.py:71
PRESETS = ("fastest", "fast", "medium", "thorough")


def main():
    if not os.path.isdir(REF):
        print("no reference tree: nothing to fetch")
        return 0
    images = []
    for s in SETS:
        for fmt in sorted(os.listdir(os.path.join(REF, s))):
            d = os.path.join(REF, s, fmt)
            if not os.path.isdir(d):
                continue
            for name in sorted(os.listdir(d)):
                stem, ext = os.path.splitext(name)
                parts = stem.split("-")
                if ext not in EXTS or len(parts) < 3 or parts[0] not in ("ldr", "ldrs", "hdr"):
                    continue
                os.makedirs(os.path.join(DST, s, fmt), exist_ok=True)
                shutil.copyfile(os.path.join(d, name), os.path.join(DST, s, fmt, name))
                flags = parts[3] if len(parts) > 3 else ""
                images.append({"set": s, "dir": fmt, "file": name, "profile": parts[0], "format": parts[1], "flags": flags,
                               "sha256": hashlib.sha256(open(os.path.join(d, name), "rb").read()).hexdigest(), "ref_psnr": {}})
    by_key = {(i["set"], i["file"]): i for i in images}
    for s in SETS:
        for preset in PRESETS:
            path = os.path.join(REF, s, "astc_reference-5.0-avx2_%s_results.csv" % preset)
            if not os.path.exists(path):
                continue
            for row in csv.DictReader(open(path)):
                img = by_key.get((row["Image Set"], row["Name"]))
                if img is not None:
                    img["ref_psnr"]["%s/%s" % (preset, row["Block Size"])] = float(row["PSNR"])
    # ... and the coding rates the reference recorded for its current main branch on its own test machine (column "Coding
    # Rate", Mtexels/s: tools/corpus_rates.py prints them next to what this library and the reference reach on this host)
    for s in SETS:
        for preset in PRESETS:
            path = os.path.join(REF, s, "astc_reference-main-avx2_%s_results.csv" % preset)
            if not os.path.exists(path):
                continue
            for row in csv.DictReader(open(path)):
                img = by_key.get((row["Image Set"], row["Name"]))
                if img is not None:
                    img.setdefault("ref_coding_rate", {})["%s/%s" % (preset, row["Block Size"])] = float(row["Coding Rate"])
    json.dump({"source": "reference Test/Images, astc_reference-5.0-avx2_*_results.csv (PSNR), astc_reference-main-avx2_*_results.csv (coding rate)", "images": images},
              open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("fetched %d images into %s" % (len(images), os.path.relpath(DST)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
