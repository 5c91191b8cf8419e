#!/usr/bin/env python3
"""Copies the reference's command line test DATA (not sources) into tests/golden/cli_data/: the three 1x1 known-answer
pairs, the 8x8 test tiles in every container the CLI reads, and the corrupt .astc headers of its negative tests
(/root/reference/Test/Data, used by /root/reference/Test/astc_test_functional.py).  tests/test_cli_functional.py runs on
the GPU box, where /root/reference does not exist; this script is how the copies were made (run it in the dev container).
A SHA-256 manifest is written next to them."""
import hashlib, json, os, shutil
SRC = "/root/reference/Test/Data"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_data")
FILES = ["LDR-A-1x1.png", "LDR-A-1x1.astc", "LDRS-A-1x1.png", "LDRS-A-1x1.astc", "HDR-A-1x1.exr", "HDR-A-1x1.astc",
         "negative_block_size.astc", "negative_huge.astc", "negative_magic.astc", "negative_overflow.astc", "negative_short.astc", "empty.unk"] + \
        ["Tiles/" + f for f in ("ldr.png", "ldr.bmp", "ldr.dds", "ldr.jpg", "ldr.ktx", "ldr.tga", "ldr_0.png", "ldr_1.png", "ldr-complex.png",
                                "ldr.astc", "hdr.exr", "hdr.hdr", "hdr-complex.exr", "hdr.astc")]
os.makedirs(os.path.join(DST, "Tiles"), exist_ok=True)
manifest = {}
for f in FILES:
    shutil.copyfile(os.path.join(SRC, f), os.path.join(DST, f))
    manifest[f] = hashlib.sha256(open(os.path.join(DST, f), "rb").read()).hexdigest()
json.dump(manifest, open(os.path.join(DST, "manifest.json"), "w"), indent=1, sort_keys=True)
print("copied %d files" % len(FILES))
