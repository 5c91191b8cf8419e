#!/bin/bash
# Build an experimental variant of the product library: tools/build_variant.sh <name> "<extra hipcc flags>" ["<flags of the two LDR fixed-context builds instead of max-ilp>"]
# -> astc-encoder_amd/variants/libastcenc_amd_<name>.so (git-ignored, travels with gpurun; deleted at round end)
set -e
cd "$(dirname "$0")/../astc-encoder_amd"
make -s -j12 OUT=variants/libastcenc_amd_$1.so OBJDIR=build/v_$1 EXTRA="$2" ${3:+ILP="$3"} variant
echo built variants/libastcenc_amd_$1.so
